"""dev helper (gpurun): median frame latency (graph sync -> solve -> frame results) of small graphs under
plan options: one isolated tile vs halo tiles at several depths.  40 frames per cell, 4 distinct graphs
round-robin, medians of sync / solve+results / total in ms."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params, default_sync_params, default_tri_params

names = sys.argv[1:] or ["g20", "g18", "g16", "tum", "g15", "g14", "g13", "v2000", "v3000"]
cfgs = [("single", dict(tile_single_max=2048)), ("auto", dict(tile_single_max=1)),
        ("d3", dict(tile_single_max=1, tile_depth=3)), ("d4", dict(tile_single_max=1, tile_depth=4)),
        ("d5", dict(tile_single_max=1, tile_depth=5)), ("d6", dict(tile_single_max=1, tile_depth=6))]
p, sp = default_params(), default_sync_params()
Kinv = np.linalg.inv(np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]])).astype(np.float32)
for name in names:
    frames = [graphgen.named(name, seed=k) for k in range(4)]
    iters = frames[0][1]
    row = []
    for label, opts in cfgs:
        r = GraphRegularizer.empty(device=0, **opts)
        ts = []
        for k in range(44):
            g = frames[k & 3][0]
            tp = default_tri_params(g.width, g.height)
            var = np.full(g.V, 1e-4, np.float32)
            t0 = time.perf_counter()
            scale = r.sync_features(g.pos, g.z, var, g.tris, sp)
            t1 = time.perf_counter()
            r.step(p, iters, sync=False)
            r.frame_results(p, Kinv, tp, scale_back=scale, with_edges=True, with_coverage=True)
            t2 = time.perf_counter()
            ts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3))
        ts = np.array(ts[4:])
        md = np.median(ts, axis=0)
        row.append("%s %.3f (%.2f+%.2f, tiles %d d%d)" % (label, md[2], md[0], md[1], r.info("num_tiles"), r.info("tile_depth")))
        r.close()
    print("V=%5d %-6s " % (frames[0][0].V, name) + " | ".join(row), flush=True)
