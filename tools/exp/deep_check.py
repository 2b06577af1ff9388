import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.util import graphgen, make_oracle, oracle_params, bits
from flame_ros_amd.regularizer import GraphRegularizer, default_params
g = graphgen.synthetic(5000, seed=1)
o = make_oracle(g); o.solve(oracle_params(), 6)
r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, path=2, tile_own=256, tile_depth=6, plan_device=0, lane_order=0)
r.step(default_params(), 6)
x, w1, w2, q = r.download()
badq = np.argwhere((bits(q) != bits(o.q)).any(axis=1)).ravel()
print("bad x", int((bits(x) != bits(o.x)).sum()), "bad q rows", len(badq))
e_o2i = r.plan_array("e_o2i", np.int32)
td = r.plan_array("tiles", np.int32).reshape(r.info("num_tiles"), -1)
emap = r.plan_array("t_emap", np.int32)
for e in badq[:12]:
    k = e_o2i[e]
    t = int(np.searchsorted(td[:, 3], k, side="right") - 1)
    D = td[t]
    loc = np.argwhere(emap[D[9]:D[9] + D[5]] == k).ravel()
    print("edge", e, "internal", k, "tile", t, "estart", D[3], "e_own", D[4], "e_loc", D[5], "local", loc.tolist(), "q", q[e], "want", o.q[e])
