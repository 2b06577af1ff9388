import os, sys, numpy as np, time
sys.path.insert(0, '/root/repo')
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer
g, _ = graphgen.named("50k")
def rg(l):
    h = l & 31
    return (0 if (h < 4 or 12 <= h < 16 or 20 <= h < 28) else 1) + 2 * (l >> 5)
RG = np.array([rg(l) for l in range(64)]); WG = np.arange(64) >> 3
r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=-1, lane_order=2)
td = r.plan_array("tiles", np.int32).reshape(r.info("num_tiles"), -1)
eij = r.plan_array("t_eij", np.uint32).reshape(-1, 2)
def true_cost(li, lj, ss, sd):
    tot = [0,0,0,0]
    for n,(idx, grp, mod) in enumerate(((li, RG, 16), (lj, RG, 16), (ss, WG, 8), (sd, WG, 8))):
        for gq in np.unique(grp):
            a = np.unique(idx[grp == gq]); tot[n] += np.bincount(a % mod, minlength=mod).max() - 1
    return tot
def surrogate(li, lj, ss, sd, with_rs):
    t = 0
    for idx, grp, mod, on in ((li, RG, 16, with_rs), (lj, RG, 16, True), (ss, WG, 8, True), (sd, WG, 8, True)):
        if not on: continue
        for gq in range(grp.max()+1):
            t += np.bincount(idx[grp == gq] % mod, minlength=mod).max() - 1
    return t
blocks = []
for D in td[::16]:
    e_loc, off, nslots = D[5], D[10], D[12]
    for b0 in range(0, e_loc - 63, 64): blocks.append((off + b0, nslots))
blocks = blocks[:60]
res = {k: [] for k in ("greedy", "true", "sur_nors", "sur_rs")}
for (o, nslots) in blocks:
    rec = eij[o:o+64]
    li, lj = (rec[:,0] & 0xffff).astype(np.int64), (rec[:,0] >> 16).astype(np.int64)
    ss0, sd0 = (rec[:,1] & 0xffff).astype(np.int64), (rec[:,1] >> 16).astype(np.int64)
    def attrs(p):
        pos = np.arange(64)
        ss = np.where(ss0[p] == 0xffff, nslots + pos, ss0[p]); sd = np.where(sd0[p] == 0xffff, nslots + pos, sd0[p])
        return li[p], lj[p], ss, sd
    res["greedy"].append(true_cost(*attrs(np.arange(64))))
    for name, fn in (("sur_nors", lambda p: surrogate(*attrs(p), False)), ("sur_rs", lambda p: surrogate(*attrs(p), True))):
        perm = np.arange(64); cur = fn(perm)
        for sweep in range(2):
            for i in range(64):
                best = (0, -1)
                for j in range(64):
                    if j == i: continue
                    p = perm.copy(); p[i], p[j] = p[j], p[i]
                    d = fn(p) - cur
                    if d < best[0]: best = (d, j)
                if best[1] >= 0:
                    j = best[1]; perm[i], perm[j] = perm[j], perm[i]; cur += best[0]
        res[name].append(true_cost(*attrs(perm)))
for k, v in res.items():
    if v: a = np.mean(np.array(v), 0); print(k, "rs %.2f rt %.2f ws %.2f wd %.2f total %.2f" % (*a, a.sum()))
