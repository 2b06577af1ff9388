"""A/B of the interior-first phase D (r05): the library named by FLAME_HIP_LIB, us per iteration (best of 10) and bits
against the oracle, over the sizes in argv (graphgen names)."""
import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params
from tests.util import make_oracle, oracle_params
p = default_params()
for name in (sys.argv[1:] or ["tum", "5k", "euroc", "50k", "v100000", "200k"]):
    g, it = graphgen.named(name)
    kw = {}
    r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, **kw)
    o = make_oracle(g); o.solve(oracle_params(), 2 * it)
    r.step(p, it); r.step(p, it)
    x, w1, w2, q = r.download()
    same = all(np.array_equal(a.view(np.uint32), np.asarray(b, np.float32).view(np.uint32)) for a, b in ((x, o.x), (w1, o.w1), (w2, o.w2), (q, o.q)))
    best = 1e9
    for _ in range(10):
        r.step(p, it); best = min(best, r.last_solve_ms()[0])
    print("%-8s %-28s depth %d resident %d: %.4f us/it  %s" % (name, os.environ.get("FLAME_HIP_LIB", "default").split("/")[-1], r.info("tile_depth"), r.info("persist_used"), best * 1e3 / it, "bit-exact" if same else "MISMATCH"), flush=True)
    r.close()
