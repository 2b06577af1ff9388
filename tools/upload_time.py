"""dev helper: wall time of re-uploads (plan build + H2D) and of a whole frame on one handle."""
import os, sys, time, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params
name = sys.argv[1] if len(sys.argv) > 1 else "50k"
opts = dict((a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[2:])  # e.g. plan_device=0 tile_own=1200
g, iters = graphgen.named(name)
r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, **opts)
p = default_params()
for k in range(5):
    t0 = time.perf_counter()
    r.reupload(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris)
    t1 = time.perf_counter()
    r.step(p, iters, sync=True)
    t2 = time.perf_counter()
    x = r.download(with_q=False)[0]
    t3 = time.perf_counter()
    print("upload %.3f ms  solve %.3f ms  download %.3f ms  total %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3), file=sys.stderr)
