"""Resident small graphs: microseconds per PD iteration with the launches per round and with ONE launch of
persistent tiles (option persist), over tile sizes / halo depths that keep the graph on <= 32 tiles."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params
p = default_params()


def us_per_it(r, iters):
    best = 1e9
    for _ in range(12):
        r.step(p, iters)
        ms, _l = r.last_solve_ms()
        best = min(best, ms)
    return best * 1e3 / iters


for name in (sys.argv[1:] or ["tum", "v800", "v2000"]):
    g, it = graphgen.named(name)
    it = 200
    base = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0)
    print("%-6s default plan: %d tiles depth %d: %.3f us/it" % (name, base.info("num_tiles"), base.info("tile_depth"), us_per_it(base, it)))
    base.close()
    for ntl in (32, 24, 16):
        own = -(-g.V // ntl)
        for depth in (3, 4, 5, 6, 8):
            res = []
            for persist in (0, 1):
                try:
                    r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, tile_own=own,
                                         tile_depth=depth, persist=persist)
                except Exception as e:
                    res.append("n/a"); continue
                t = us_per_it(r, it)
                res.append("%.3f%s" % (t, "" if r.info("persist_used") == persist else "!"))
                nt, thr, ept = r.info("num_tiles"), r.info("tile_threads"), r.info("tile_ept")
                r.close()
            print("   own %3d depth %d (%2d tiles, %4d threads, ept %d): launches %s  persist %s us/it" % (own, depth, nt, thr, ept, res[0], res[1]))
