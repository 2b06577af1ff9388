"""Frames below the isolated-tile threshold (<= 896 vertices): one isolated tile (the facade's choice so far)
against persistent halo tiles (tile_single_max lowered, persist = 2)."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params, default_sync_params, default_tri_params
p, sp = default_params(), default_sync_params()
Kinv = np.linalg.inv(np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]])).astype(np.float32)
for name in sys.argv[1:] or ["v300", "v500", "v700", "v850"]:
    frames = [graphgen.named(name, seed=k) for k in range(4)]
    for label, kw in (("single tile", dict(tile_single_max=896, stream_depth=5, persist=2)),
                      ("persistent ", dict(tile_single_max=128, stream_depth=5, persist=2))):
        r = GraphRegularizer.empty(device=0, **kw)
        tt = []
        for k in range(40):
            g = frames[k & 3][0]
            tp = default_tri_params(g.width, g.height)
            var = np.full(g.V, 1e-4, np.float32)
            t0 = time.perf_counter()
            scale = r.sync_features(g.pos, g.z, var, g.tris, sp)
            r.step(p, 200, sync=False)
            out = r.frame_results(p, Kinv, tp, scale_back=scale, with_edges=True, with_coverage=True)
            if k >= 8: tt.append((time.perf_counter() - t0) * 1e3)
        print("%s %s: frame p50 %.3f ms  (tiles %d depth %d persist_used %d mini %d on_device %d)" % (
            name, label, np.median(tt), r.info("num_tiles"), r.info("tile_depth"), r.info("persist_used"), r.info("plan_mini"), r.info("plan_on_device")))
        r.close()
