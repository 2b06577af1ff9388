"""dev helper (gpurun): device time of the iterations of ONE solve, launched directly (the first solve of a
plan) against the same solve replayed from a hipGraph (third solve) -- what a frame stream would gain if
its first solve could replay an instantiated graph."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params
p = default_params()
for name, opts in (("tum", dict(tile_single_max=1, tile_depth=5)), ("5k", {}), ("euroc", {}), ("50k", {})):
    g, iters = graphgen.named(name)
    first, replay = [], []
    for rep in range(40):
        r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, lane_order=0, **opts)
        r.step(p, iters); first.append(r.last_solve_ms()[0])
        r.step(p, iters); r.step(p, iters); replay.append(r.last_solve_ms()[0])
        n = r.last_solve_ms()[1]
        r.close()
    first, replay = first[10:], replay[10:]  # (clocks settled)
    print("%-6s launches %3d  direct %.4f ms  replay %.4f ms  (%.2f vs %.2f us per launch)" % (
        name, n, np.median(first), np.median(replay), np.median(first) * 1e3 / n, np.median(replay) * 1e3 / n), flush=True)
