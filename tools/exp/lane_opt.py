"""dev: how low can the modelled phase-D conflict cycles go?  Greedy variants + swap refinement."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer
g, _ = graphgen.named("50k")
def rg(l):
    h = l & 31
    return (0 if (h < 4 or 12 <= h < 16 or 20 <= h < 28) else 1) + 2 * (l >> 5)
RG = [rg(l) for l in range(64)]
r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=-1, lane_order=0)
td = r.plan_array("tiles", np.int32).reshape(r.info("num_tiles"), -1)
eij = r.plan_array("t_eij", np.uint32).reshape(-1, 2)

def cost_of(assign, E):  # assign[lane] = edge index or -1 ; E = list of (li, lj, ss, sd)
    tot = 0
    for kind in range(4):
        groups = {}
        for lane, e in enumerate(assign):
            if e < 0: continue
            li, lj, ss, sd = E[e]
            if kind == 0: key, a, mod = RG[lane], li, 16
            elif kind == 1: key, a, mod = RG[lane], lj, 16
            elif kind == 2: key, a, mod = lane >> 3, (ss if ss != 0xffff else 100000 + lane), 8
            else: key, a, mod = lane >> 3, (sd if sd != 0xffff else 100000 + lane), 8
            groups.setdefault(key, set()).add(a)
        for s in groups.values():
            occ = np.bincount(np.array(list(s)) % mod, minlength=mod).max()
            tot += occ - 1
    return tot

def greedy(E, order):
    c = len(E); assign = [-1] * c
    for e in order:
        best, bc = -1, 1 << 30
        for lane in range(c):
            if assign[lane] >= 0: continue
            assign[lane] = e; cst = cost_of(assign, E); assign[lane] = -1
            if cst < bc: bc, best = cst, lane
        assign[best] = e
    return assign

def swaps(assign, E, passes=2):
    c = len(assign); cur = cost_of(assign, E)
    for _ in range(passes):
        improved = False
        for a in range(c):
            for b in range(a + 1, c):
                assign[a], assign[b] = assign[b], assign[a]
                n = cost_of(assign, E)
                if n < cur: cur = n; improved = True
                else: assign[a], assign[b] = assign[b], assign[a]
        if not improved: break
    return assign

rng = np.random.default_rng(0)
res = {"sorted": [], "greedy_exact": [], "greedy_rev": [], "greedy_hard_first": [], "greedy_rand": []}
tiles = rng.choice(len(td), 6, replace=False)
for t in tiles:
    D = td[t]; e_loc, off = D[5], D[10]
    for b0 in list(range(0, e_loc, 64))[:6]:
        rec = eij[off + b0: off + min(b0 + 64, e_loc)]
        E = [(int(x & 0xffff), int(x >> 16), int(y & 0xffff), int(y >> 16)) for x, y in rec]
        c = len(E)
        res["sorted"].append(cost_of(list(range(c)), E))
        a = greedy(E, range(c)); res["greedy_exact"].append(cost_of(a, E))
        a = greedy(E, range(c - 1, -1, -1)); res["greedy_rev"].append(cost_of(a, E))
        from collections import Counter
        cs = Counter(e[2] % 8 for e in E if e[2] != 0xffff); cd = Counter(e[3] % 8 for e in E if e[3] != 0xffff); ct = Counter(e[1] % 16 for e in E)
        hard = sorted(range(c), key=lambda i: -(cs.get(E[i][2] % 8, 0) + cd.get(E[i][3] % 8, 0) + ct[E[i][1] % 16]))
        a = greedy(E, hard); res["greedy_hard_first"].append(cost_of(a, E))
        a = greedy(E, list(rng.permutation(c))); res["greedy_rand"].append(cost_of(a, E))
for k, v in res.items(): print(k, "mean extra cycles/block %.2f" % np.mean(v), "n", len(v))
