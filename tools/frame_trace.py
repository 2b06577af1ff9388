"""dev helper: one FLaME frame through the library the way flame::Flame::updateGraph drives it
(graph sync -> solve -> frame_results with coverage), N times; used under rocprofv3 by
tools/prof_frame_trace.sh."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params, default_sync_params, default_tri_params
name = sys.argv[1] if len(sys.argv) > 1 else "50k"
opts = {k: int(v) for k, v in (a.split("=") for a in sys.argv[2:])}
frames = [graphgen.named(name, seed=k) for k in range(2)]
iters = frames[0][1]
r = GraphRegularizer.empty(device=0, **opts)
p, sp = default_params(), default_sync_params()
Kinv = np.linalg.inv(np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]])).astype(np.float32)
for k in range(8):
    g = frames[k & 1][0]
    tp = default_tri_params(g.width, g.height)
    var = np.full(g.V, 1e-4, np.float32)
    t0 = time.perf_counter()
    scale = r.sync_features(g.pos, g.z, var, g.tris, sp)
    t1 = time.perf_counter()
    r.step(p, iters, sync=False)
    out = r.frame_results(p, Kinv, tp, scale_back=scale, with_edges=True, with_coverage=True)
    t2 = time.perf_counter()
    print("sync %.3f  solve+results %.3f  total %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3), file=sys.stderr)
