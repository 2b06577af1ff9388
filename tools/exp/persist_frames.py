"""Frame stream of small graphs (graph sync -> 200 iterations -> results) with the launches per round and with
persistent tiles (option persist, ~24 tiles): frame latency and device time of the solve."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params, default_sync_params, default_tri_params
name = sys.argv[1] if len(sys.argv) > 1 else "tum"
frames = [graphgen.named(name, seed=k) for k in range(4)]
p, sp = default_params(), default_sync_params()
Kinv = np.linalg.inv(np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]])).astype(np.float32)
for persist in (0, 1, 0, 1):
    kw = dict(tile_own=max(32, -(-frames[0][0].V // 24)), tile_depth=5) if persist else {}  # ~24 tiles on one XCD
    r = GraphRegularizer.empty(device=0, tile_single_max=896, stream_depth=5, persist=persist, **kw)
    tt, dv = [], []
    for k in range(40):
        g = frames[k & 3][0]
        tp = default_tri_params(g.width, g.height)
        var = np.full(g.V, 1e-4, np.float32)
        t0 = time.perf_counter()
        scale = r.sync_features(g.pos, g.z, var, g.tris, sp)
        r.step(p, 200, sync=False)
        out = r.frame_results(p, Kinv, tp, scale_back=scale, with_edges=True, with_coverage=True)
        t2 = time.perf_counter()
        if k >= 8: tt.append((t2 - t0) * 1e3); dv.append(r.last_solve_ms()[0])
    print("persist %d: frame p50 %.3f ms, device solve p50 %.3f (min %.3f max %.3f)  used %d tiles %d depth %d mini %d reused %d" % (
        persist, np.median(tt), np.median(dv), min(dv), max(dv), r.info("persist_used"), r.info("num_tiles"), r.info("tile_depth"), r.info("plan_mini"), r.info("plan_reused")))
    r.close()
