"""dev helper: wall time of one FLaME frame through the library the way flame::Flame::updateGraph
drives it: graph sync (features + triangulation in) -> solve -> costs -> download -> triangle stage."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params, default_sync_params, default_tri_params
name = sys.argv[1] if len(sys.argv) > 1 else "50k"
g, iters = graphgen.named(name)
var = np.full(g.V, 1e-4, np.float32)
opts = {k: int(v) for k, v in (a.split("=") for a in sys.argv[2:])}  # e.g. tile_single_max=2048 (the facade's setting)
r = GraphRegularizer.empty(device=0, **opts)
p, sp, tp = default_params(), default_sync_params(), default_tri_params(g.width, g.height)
Kinv = np.linalg.inv(np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]])).astype(np.float32)
for k in range(6):
    t0 = time.perf_counter()
    r.sync_features(g.pos, g.z, var, g.tris, sp)
    t1 = time.perf_counter()
    r.step(p, iters, sync=True)
    t2 = time.perf_counter()
    c = r.costs(p)
    x = r.download(with_q=False)[0]
    e = r.edges()
    t3 = time.perf_counter()
    tn, tv, vn = r.triangles(Kinv, tp)
    t4 = time.perf_counter()
    print("sync %.3f  solve %.3f  costs+x+edges %.3f  triangles %.3f  total %.3f ms" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3), file=sys.stderr)
for k in range(6):  # same frame, results through the one-synchronisation call the façade uses
    t0 = time.perf_counter()
    scale = r.sync_features(g.pos, g.z, var, g.tris, sp)
    t1 = time.perf_counter()
    r.step(p, iters, sync=False)
    out = r.frame_results(p, Kinv, tp, scale_back=scale, with_edges=True)
    t2 = time.perf_counter()
    print("sync %.3f  solve+frame_results %.3f  total %.3f ms" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3), file=sys.stderr)
