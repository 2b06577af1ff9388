"""Dev aid: the round split of EVERY resident tile (option persist_prof, one solve per tile) beside the tile's size:
who is the slowest tile of a round, and by how much.  python tools/exp/tile_round_profile.py [name] [--opt k=v]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flame_ros_amd import graphgen  # noqa: E402
from flame_ros_amd.regularizer import GraphRegularizer, default_params  # noqa: E402

args, name, kw = sys.argv[1:], "50k", {}
i = 0
while i < len(args):
    if args[i] == "--opt":
        k, v = args[i + 1].split("=")
        kw[k] = int(v)
        i += 2
    else:
        name = args[i]
        i += 1
g, it = graphgen.named(name)
p = default_params()
with GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, **kw) as r:
    r.step(p, it)
    r.step(p, it)
    ms = []
    for _ in range(6):
        r.step(p, it)
        ms.append(r.last_solve_ms()[0])
    nt = r.info("num_tiles")
    t = r.plan_array("tiles", np.int32).reshape(-1, 47)
    rows = []
    for tile in range(nt):
        r.set_option("persist_prof", tile + 1)
        r.step(p, it)
        v = [r.info("persist_prof_%d" % k) for k in range(5)]
        n = max(v[3] - 1, 1)
        rows.append([v[0] / n / 100.0, v[1] / n / 100.0, v[2] / n / 100.0, v[4] / n])
    rows = np.asarray(rows)
    cost = t[:, 5] + 2 * t[:, 2]
    print("%s: %d tiles depth %d, %.4f us/it (best of 6); imbalance info %d%%" % (name, nt, r.info("tile_depth"), min(ms) * 1e3 / it, r.info("tile_imbalance_pct")))
    for nm, col in (("iterate+store", rows[:, 0]), ("poll", rows[:, 1]), ("apply+barrier", rows[:, 2]), ("round", rows[:, :3].sum(1))):
        print("  %-14s min %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f us" % (nm, col.min(), np.percentile(col, 10), np.median(col), np.percentile(col, 90), col.max()))
    order = np.argsort(-rows[:, 0])[:8]
    print("  slowest tiles by iterate: " + ", ".join("t%d %.2f (n_own %d n_ext %d e_loc %d cost %d)" % (k, rows[k, 0], t[k, 1], t[k, 2], t[k, 5], cost[k]) for k in order))
    print("  cost model e_loc + 2 n_ext: max/mean %.3f; corr(iterate, cost) %.3f; iterate max/median %.3f" % (
        cost.max() / cost.mean(), np.corrcoef(rows[:, 0], cost)[0, 1], rows[:, 0].max() / np.median(rows[:, 0])))
    # least-squares fit iterate ~ a n_ext + b e_loc + c
    A = np.stack([t[:, 2], t[:, 5], np.ones(nt)], 1).astype(np.float64)
    coef, *_ = np.linalg.lstsq(A, rows[:, 0], rcond=None)
    print("  fit iterate_us = %.5f n_ext + %.5f e_loc + %.3f (residual rms %.3f)" % (coef[0], coef[1], coef[2], np.sqrt(np.mean((A @ coef - rows[:, 0]) ** 2))))
    # ... with the incidence slots (phase P walks every row to its 64-group's pitch) and the updated vertices
    for cols, names in (((12, 5), "nslots e_loc"), ((12, 5, 6), "nslots e_loc n_upd"), ((12,), "nslots")):
        A = np.stack([t[:, c] for c in cols] + [np.ones(nt)], 1).astype(np.float64)
        coef, *_ = np.linalg.lstsq(A, rows[:, 0], rcond=None)
        pred = A @ coef
        print("  fit iterate_us ~ [%s, 1] = %s (residual rms %.3f, corr %.3f, pred max/mean %.3f)" % (
            names, " ".join("%.5f" % c for c in coef), np.sqrt(np.mean((pred - rows[:, 0]) ** 2)), np.corrcoef(pred, rows[:, 0])[0, 1], pred.max() / pred.mean()))
    np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gpurun_out", "tile_rounds_%s.npy" % name), np.concatenate([t[:, :13], (rows * 1000).astype(np.int64)], 1))
