"""How much the lane order (greedy against LDS bank conflicts, modelled on the 16-byte slots' ds_write_b96) is worth per size --
and so what a model of the 12-byte layout's stores could still find on the fat tiles: lane_order 0 (sorted edges) vs 1."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params
p = default_params()
for name in (sys.argv[1:] or ["50k", "v100000", "v160000", "200k"]):
    g, it = graphgen.named(name)
    for lo in (0, 1, 0, 1):
        r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, lane_order=lo)
        best = 1e9
        for _ in range(10):
            r.step(p, it); best = min(best, r.last_solve_ms()[0])
        print("%-8s lane_order %d slot12 %d: %.4f us/it" % (name, lo, r.info("tile_slot12"), best * 1e3 / it), flush=True)
        r.close()
