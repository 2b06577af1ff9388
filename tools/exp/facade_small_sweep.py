"""flame::Flame::update medians of small frames against the tile size the facade's `persist = 2` rule picks
(FLAME_HIP_PERSIST_OWN / FLAME_HIP_PERSIST_VMAX override the 50 own vertices / 1280-vertex limit)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools import facade_bench
for w in (sys.argv[1:] or ["v800", "tum", "v1600", "v2000", "v3000"]):
    row = []
    for own in (0, 24, 32, 40, 50, 64):
        env = {"FLAME_HIP_PERSIST_VMAX": "4000", "FLAME_HIP_PERSIST_OWN": str(own)} if own else {"FLAME_HIP_PERSIST_VMAX": "0"}
        r = facade_bench.run(w, repeats=15, getters=0, env=env)
        row.append("own %2d: %.3f (sync %.3f solve %.3f)" % (own, r["update_ms"]["p50"], r["sync_graph_ms_p50"], r["nltgv2_ms_p50"]))
    print(w, " | ".join(row), flush=True)
