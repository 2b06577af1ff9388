"""r06 experiment: up to 64 resident tiles on ONE XCD (two per CU) -- frames of 1.3 k .. 2.5 k vertices through the L2 hand-off.
Run with a library built with kOneXcdTiles = 64 (plan.h) against one_xcd = 0 on the same library."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params
from tests.util import make_oracle, oracle_params
p = default_params()
for rep in range(2):
    for V in (1200, 1400, 1600, 2000, 2400, 2560):
        for kw in (dict(one_xcd=0), dict()):
            g, it = graphgen.synthetic(V, seed=2), 200
            o = make_oracle(g); o.solve(oracle_params(), 2 * it)
            r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, **kw)
            r.step(p, it); r.step(p, it)
            x, w1, w2, q = r.download()
            same = all(np.array_equal(a.view(np.uint32), np.asarray(b, np.float32).view(np.uint32)) for a, b in ((x, o.x), (w1, o.w1), (w2, o.w2), (q, o.q)))
            best = 1e9
            for _ in range(12):
                r.step(p, it); best = min(best, r.last_solve_ms()[0])
            print("V %5d %-14s: %.4f us/it  tiles %d depth %d one XCD %d gave_up %d recovered %d %s" % (V, kw, best * 1e3 / it, r.info("num_tiles"), r.info("tile_depth"),
                  r.info("one_xcd_used"), r.info("persist_gave_up"), r.info("persist_recovered"), "bit-exact" if same else "MISMATCH"), flush=True)
            r.close()
