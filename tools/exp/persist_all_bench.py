"""Resident graphs of 33..256 tiles: launches per round against ONE launch of tiles persistent over all XCDs
(option persist = 3 of tools/exp/persist_all_xcd.patch -- apply it first: state arrays in uncached device memory).  us per iteration, and the bits against the oracle."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params
from tests.util import make_oracle, oracle_params
p = default_params()
for name in (sys.argv[1:] or ["v2000", "5k", "euroc", "50k"]):
    g, it = graphgen.named(name)
    o = make_oracle(g); o.solve(oracle_params(), it)
    for persist in (0, 3):
        r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, persist=persist)
        r.step(p, it)
        x, w1, w2, q = r.download()
        ok = np.array_equal(x.view(np.uint32), o.x.view(np.uint32)) and np.array_equal(q.view(np.uint32), o.q.view(np.uint32))
        best = 1e9
        for _ in range(8):
            r.step(p, it)
            best = min(best, r.last_solve_ms()[0])
        print("%-6s persist %d: %.3f us/it (%.3f ms per %d)  tiles %d depth %d used %d  bit-exact after the first solve: %s" % (
            name, persist, best * 1e3 / it, best, it, r.info("num_tiles"), r.info("tile_depth"), r.info("persist_used"), ok))
        r.close()
