"""dev: modelled LDS conflict cycles of phase D per 64-edge block (MI355X guide lane groups),
from a host-built plan (no GPU needed)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer
g, _ = graphgen.named(sys.argv[1] if len(sys.argv) > 1 else "50k")
def rg(l):
    h = l & 31
    return (0 if (h < 4 or 12 <= h < 16 or 20 <= h < 28) else 1) + 2 * (l >> 5)
RG = np.array([rg(l) for l in range(64)])
for lo in (0, 2):
    r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=-1, lane_order=lo)
    td = r.plan_array("tiles", np.int32).reshape(r.info("num_tiles"), -1)
    eij = r.plan_array("t_eij", np.uint32).reshape(-1, 2)
    tot = dict(rs=0, rt=0, ws=0, wd=0); nb = 0
    for D in td:
        e_loc, off, nslots = D[5], D[10], D[12]
        for b0 in range(0, e_loc, 64):
            rec = eij[off + b0: off + min(b0 + 64, e_loc)]
            c = len(rec); nb += 1
            li, lj = rec[:, 0] & 0xffff, rec[:, 0] >> 16
            ss, sd = rec[:, 1] & 0xffff, rec[:, 1] >> 16
            lanes = np.arange(c)
            ss = np.where(ss == 0xffff, nslots + lanes, ss); sd = np.where(sd == 0xffff, nslots + lanes, sd)
            for name, idx, grp, mod in (("rs", li, RG[:c], 16), ("rt", lj, RG[:c], 16), ("ws", ss, lanes >> 3, 8), ("wd", sd, lanes >> 3, 8)):
                for gq in np.unique(grp):
                    a = np.unique(idx[grp == gq])           # distinct addresses (equal ones broadcast)
                    occ = np.bincount(a % mod, minlength=mod).max()
                    tot[name] += occ - 1
    print("lane_order", lo, "blocks", nb, {k: round(v / nb, 2) for k, v in tot.items()}, "extra cycles/block", round(sum(tot.values()) / nb, 2))
    r.close()
