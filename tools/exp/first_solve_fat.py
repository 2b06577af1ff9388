"""r05: the FIRST solve of a plan (poll lists in local order, sorted local edges) against later ones, on fat tiles."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params
p = default_params()
for V in [int(a) for a in (sys.argv[1:] or ["60000", "110000", "150000", "180000", "195316", "200000", "215000"])]:
    for seed in (V, 1000 + 395):
        g = graphgen.synthetic(V, seed=seed)
        r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0)
        ts = []
        for k in range(4):
            r.step(p, 60)
            ts.append(r.last_solve_ms()[0])
        print("V %6d seed %6d: tiles %d depth %d cfg %d/%d slot12 %d lds %d: solves of 60 iterations %s ms, wait_max %d us, resident %d" % (
            V, seed, r.info("num_tiles"), r.info("tile_depth"), r.info("tile_threads"), r.info("tile_ept"), r.info("tile_slot12"), r.info("tile_lds_bytes"),
            " ".join("%.3f" % t for t in ts), r.info("persist_wait_us_max"), r.info("persist_used")), flush=True)
        r.close()
