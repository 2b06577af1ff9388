import sys, numpy as np
sys.path.insert(0, '/root/repo')
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params
p = default_params()
g, _ = graphgen.named("tum")
for own, depth in ((75, 5), (50, 5), (38, 5), (50, 4)):
    r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, tile_own=own, tile_depth=depth, persist=1, persist_prof=1)
    for _ in range(3): r.step(p, 200)
    ms, _l = r.last_solve_ms()
    v = [r.info("persist_prof_%d" % i) for i in range(5)]
    n = max(v[4] - 1, 1)
    print("own %d depth %d tiles %d: %.3f us/it; per round (us): iterate+store issue %.2f, store ack+barrier %.2f, flags %.2f, re-read %.2f  (%d rounds)" % (
        own, depth, r.info("num_tiles"), ms * 1e3 / 200, v[0] * 0.01 / n, v[1] * 0.01 / n, v[2] * 0.01 / n, v[3] * 0.01 / n, v[4]))
    r.close()
