import sys; sys.path.insert(0, '/root/repo')
from tools import facade_bench
facade_bench.build(force=True)
for w in ("tum", "5k", "euroc", "50k"):
    f = facade_bench.run(w, repeats=5 if w != "50k" else 3, getters=1, env={"FLAME_BENCH_FRONTEND": "1"})
    print(w, "from_features update p50 %.3f ms, triangulate p50 %.3f ms" % (f["update_ms"]["p50"], f["triangulate_ms_p50"]))
