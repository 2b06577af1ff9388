"""Resident small graphs (r05): us per PD iteration over tile size x halo depth (best of 12 solves of 200 iterations)."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params
p = default_params()
for name in (sys.argv[1:] or ["5k", "euroc", "tum", "v2500", "v20000"]):
    g, it = graphgen.named(name)
    it = 200
    rows = []
    for own in (0, 24, 32, 40, 48, 64, 80, 100, 128, 160):
        for depth in ((0,) if own == 0 else (3, 4, 5, 6)):
            try:
                r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, tile_own=own, tile_depth=depth)
            except Exception as e:
                continue
            best = 1e9
            for _ in range(12):
                r.step(p, it); best = min(best, r.last_solve_ms()[0])
            rows.append((best * 1e3 / it, own, depth, r.info("num_tiles"), r.info("tile_depth"), r.info("tile_threads"), r.info("tile_ept"), r.info("persist_used")))
            r.close()
    rows.sort()
    print(name, "V", g.V)
    for us, own, depth, nt, d, th, ept, res in rows[:8] + [x for x in rows if x[1] == 0]:
        print("   %.4f us/it  own %3s depth %s -> %3d tiles depth %d %4d x %d resident %d" % (us, own or "auto", depth or "auto", nt, d, th, ept, res), flush=True)
