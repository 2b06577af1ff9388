import os, sys, numpy as np, time
sys.path.insert(0, '/root/repo')
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer
g, _ = graphgen.named("50k")
def rg(l):
    h = l & 31
    return (0 if (h < 4 or 12 <= h < 16 or 20 <= h < 28) else 1) + 2 * (l >> 5)
RG = np.array([rg(l) for l in range(64)])
WG = np.arange(64) >> 3
r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=-1, lane_order=2)
td = r.plan_array("tiles", np.int32).reshape(r.info("num_tiles"), -1)
eij = r.plan_array("t_eij", np.uint32).reshape(-1, 2)
def true_cost(li, lj, ss, sd):
    c = len(li); tot = [0,0,0,0]
    for n,(idx, grp, mod) in enumerate(((li, RG[:c], 16), (lj, RG[:c], 16), (ss, WG[:c], 8), (sd, WG[:c], 8))):
        for gq in np.unique(grp):
            a = np.unique(idx[grp == gq])
            tot[n] += np.bincount(a % mod, minlength=mod).max() - 1
    return tot
def cost_vec(li, lj, ss, sd):
    return sum(true_cost(li, lj, ss, sd))
rng = np.random.default_rng(0)
blocks = []
for D in td[::8]:
    e_loc, off, nslots = D[5], D[10], D[12]
    for b0 in range(0, e_loc - 63, 64):
        blocks.append((off + b0, nslots))
blocks = blocks[:200]
base = []; imp = []; imp2=[]
t0=time.time()
for (o, nslots) in blocks:
    rec = eij[o:o+64]
    li, lj = (rec[:,0] & 0xffff).astype(np.int64), (rec[:,0] >> 16).astype(np.int64)
    ss, sd = (rec[:,1] & 0xffff).astype(np.int64), (rec[:,1] >> 16).astype(np.int64)
    # (trash slots depend on lane; ignore: treat 0xffff as unique large ids)
    big = 100000 + np.arange(64)
    ss = np.where(ss == 0xffff, big, ss); sd = np.where(sd == 0xffff, big, sd)
    perm = np.arange(64)  # perm[lane] = edge index
    def cost_of(p):
        return cost_vec(li[p], lj[p], ss[p], sd[p])
    c0 = cost_of(perm); base.append(c0)
    # hill climbing with best-improvement per i (true cost; slow but a prototype)
    cur = c0
    for sweep in range(3):
        improved = False
        for i in range(64):
            best = (0, -1)
            for j in range(i+1, 64):
                p = perm.copy(); p[i], p[j] = p[j], p[i]
                d = cost_of(p) - cur
                if d < best[0]: best = (d, j)
            if best[1] >= 0:
                j = best[1]; perm[i], perm[j] = perm[j], perm[i]; cur += best[0]; improved = True
        if sweep == 0: imp.append(cur)
        if not improved: break
    imp2.append(cur)
    if len(imp2) % 20 == 0: print(len(imp2), np.mean(base), np.mean(imp), np.mean(imp2), time.time()-t0, flush=True)
print("greedy %.2f  after 1 sweep %.2f  converged %.2f extra cycles per block" % (np.mean(base), np.mean(imp), np.mean(imp2)))
