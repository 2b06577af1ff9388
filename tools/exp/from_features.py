"""update() from features alone (track -> gate -> built-in triangulation -> GPU tail): the built-in triangulation on the
GPU (flame_hip_delaunay, Params::triangulate_on_gpu = true, the default) against the host pool (false)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools import facade_bench
facade_bench.build(force=True)
for w in ("tum", "5k", "euroc", "50k"):
    for gpu in ("1", "0"):
        f = facade_bench.run(w, repeats=5 if w != "50k" else 3, getters=1, env={"FLAME_BENCH_FRONTEND": "1", "FLAME_BENCH_TRI_GPU": gpu})
        print(w, "triangulation on the %s: from_features update p50 %.3f ms, triangulate p50 %.3f ms" % ("GPU " if gpu == "1" else "host", f["update_ms"]["p50"], f["triangulate_ms_p50"]), flush=True)
