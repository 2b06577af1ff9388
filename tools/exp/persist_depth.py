"""Halo depth of frames on persistent tiles (the facade's options with stream_depth 4..7): 4-6 are equal within the noise
(1.2 k: 0.40-0.43 ms), 7 loses (0.45)."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params, default_sync_params, default_tri_params
p, sp = default_params(), default_sync_params()
Kinv = np.linalg.inv(np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]])).astype(np.float32)
for name in ("tum", "v900"):
    frames = [graphgen.named(name, seed=k) for k in range(4)]
    for rep in range(2):
        for d in (4, 5, 6, 7):
            r = GraphRegularizer.empty(device=0, tile_single_max=640, stream_depth=d, persist=2)
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
            print("%s depth %d: frame p50 %.3f ms (tiles %d, persist %d, threads %d ept %d)" % (name, d, np.median(tt), r.info("num_tiles"), r.info("persist_used"), r.info("tile_threads"), r.info("tile_ept")))
            r.close()
