"""What each RECALLED choice is worth against the 1e-4 RMS tolerance.

Parity with upstream is unpinned: the oracle restates the published algorithm and a handful of
choices are recalled, not read.  Whoever brings an upstream dump (tools/pin_upstream/README.md) should know which
disagreements matter: this script runs the oracle (CPU; test infrastructure) on the TUM- and EuRoC-shaped graphs of
BASELINE configs 1 / 3 with the defaults and with ONE choice flipped at a time and prints the change of the idepths
after the frame's iterations -- RMS and max over vertices, against the north_star tolerance 1e-4 RMS -- and, for the
filter thresholds of the triangle stage, how many validity flags flip.

    python tools/pin_upstream/price_list.py [--markdown] > profiles/r04_unpinned_price_list.md
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flame_ros_amd import graphgen  # noqa: E402
from oracle import COracle  # noqa: E402
from oracle.cbind import SyncParams, TriParams, default_params, graph_sync, triangles  # noqa: E402

TOL = 1e-4


def solve(g, sync=None, params=None, iters=200, var=None, pred=None, edge_perm=None):
    """Oracle graph sync + iterations on the features of g; returns x (caller's vertex order)."""
    sp = sync or SyncParams(0, 0, 1, 0.01, 0, 0.0, 0.0)
    var = np.full(g.V, 1e-4, np.float32) if var is None else var
    s = graph_sync(sp, g.pos, g.z, var, g.tris, pred)
    edges, alpha, beta = s["edges"], s["alpha"], s["beta"]
    if edge_perm is not None:  # another summation order of the -tau K^T q terms: the edge list relabelled
        p = edge_perm(len(edges))
        edges, alpha, beta = edges[p], alpha[p], beta[p]
    o = COracle(g.pos, edges, alpha, beta, s["z"], s["wgt"], x0=s["x0"])
    o.solve(params or default_params(), iters)
    return o.x.copy() * (s["scale"] if sp.rescale_data else 1.0), s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--markdown", action="store_true")
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    rows = []
    for name in ("tum", "euroc", "5k"):
        g, iters = graphgen.named(name)
        base, s0 = solve(g, iters=iters)

        def price(label, x, note=""):
            d = x.astype(np.float64) - base
            rows.append((name, label, float(np.sqrt(np.mean(d * d))), float(np.abs(d).max()), note))

        # ---- graph sync (row a7) ----
        for rule, lab in ((1, "edge weights alpha = beta = 1 (rule 1)"), (2, "alpha = 1/len, beta = 1 (rule 2)"),
                          (3, "alpha = 1, beta = 1/len (rule 3)")):
            price(lab, solve(g, SyncParams(0, 0, 1, 0.01, rule, 0.0, 0.0), iters=iters)[0])
        for ga, gb in ((2.0, 1.0), (1.0, 2.0), (0.5, 0.5), (1.1, 1.1)):
            price("gains alpha x %.1f, beta x %.1f on 1/len" % (ga, gb), solve(g, SyncParams(0, 0, 1, 0.01, 0, ga, gb), iters=iters)[0])
        var = (1e-4 * (1.0 + 4.0 * rng.random(g.V))).astype(np.float32)
        price("adaptive_data_weights on (weights 1/var, var in [1e-4, 5e-4])", solve(g, SyncParams(1, 0, 1, 0.01, 0, 0, 0), iters=iters, var=var)[0],
              "YAML default off (cfg/flame_offline_tum.yaml:89)")
        price("rescale_data on (data scaled to mean 1, state scaled back)", solve(g, SyncParams(0, 1, 1, 0.01, 0, 0, 0), iters=iters)[0],
              "YAML default off (:90)")
        pred = (g.z * (1.0 + 0.05 * rng.standard_normal(g.V))).astype(np.float32)
        price("init_with_prediction with a 5 % noisy prediction vs x0 = z", solve(g, iters=iters, pred=pred)[0], "x0 only: the fixed point is the same")
        # ---- solver (rows a2-a5) ----
        price("idepth clamp [0, 10] -> none", solve(g, params=default_params(x_min=-1e30, x_max=1e30), iters=iters)[0], "recalled default 0..10")
        price("idepth clamp [0, 10] -> [0.01, 2]", solve(g, params=default_params(x_min=0.01, x_max=2.0), iters=iters)[0])
        for it in (iters // 2, iters - 20, iters + 20, 2 * iters):
            price("iterations %d -> %d" % (iters, it), solve(g, iters=it)[0], "not a YAML key: upstream default unknown")
        price("theta 0.25 -> 1.0 (textbook Chambolle-Pock)", solve(g, params=default_params(theta=1.0), iters=iters)[0], "YAML pins 0.25 (:96)")
        price("summation order of -tau K^T q: edges relabelled at random", solve(g, iters=iters, edge_perm=lambda n: rng.permutation(n))[0],
              "Boost's out-edge order instead of ascending edge id")
        price("summation order: edge list reversed", solve(g, iters=iters, edge_perm=lambda n: np.arange(n)[::-1])[0])
        # d -> -d: exactly no change of x (K10); shown for completeness through flipped edge orientation instead, which
        # DOES change K1 (it uses w of the source, K7)
        sp = SyncParams(0, 0, 1, 0.01, 0, 0.0, 0.0)
        s = graph_sync(sp, g.pos, g.z, np.full(g.V, 1e-4, np.float32), g.tris, None)
        o = COracle(g.pos, s["edges"][:, ::-1].copy(), s["alpha"], s["beta"], s["z"], s["wgt"], x0=s["x0"])
        o.solve(default_params(), iters)
        price("edge orientation i<j -> j<i (K1 uses w of the SOURCE)", o.x, "orientation rule of graph sync is recalled")
        rows.append((name, "sign of the edge vector d (option d_sign)", 0.0, 0.0, "provably none: K10"))
        # ---- the triangulation in front (row f3's first leg): which diagonal cuts a cocircular cell.  Integer-pixel features
        # (what a detector produces) are full of cocircular quadruples; the built-in triangulators cut them by their own
        # rules (GPU: a fan from the smallest vertex id; host: the divide-and-conquer's), upstream's Triangle by another.
        # Two valid Delaunay triangulations of the SAME integer-pixel features: SciPy's on the points as they are and on
        # the points moved by < 1e-3 pixel (every tie broken at random) ----
        if name != "5k":
            from scipy.spatial import Delaunay
            pix = np.round(g.pos).astype(np.float32)
            pix = pix[np.unique(pix, axis=0, return_index=True)[1]]
            zz = (0.5 + 0.001 * pix[:, 0]).astype(np.float32) + (0.02 * rng.standard_normal(len(pix))).astype(np.float32)

            def run_on(points_for_ties):
                t = Delaunay(points_for_ties.astype(np.float64)).simplices.astype(np.int32)
                P = pix.astype(np.float64)
                d = (P[t[:, 1], 0] - P[t[:, 0], 0]) * (P[t[:, 2], 1] - P[t[:, 0], 1]) - (P[t[:, 1], 1] - P[t[:, 0], 1]) * (P[t[:, 2], 0] - P[t[:, 0], 0])
                t = t[d != 0]
                s_ = graph_sync(SyncParams(0, 0, 1, 0.01, 0, 0.0, 0.0), pix, zz, np.full(len(pix), 1e-4, np.float32), t, None)
                o_ = COracle(pix, s_["edges"], s_["alpha"], s_["beta"], s_["z"], s_["wgt"], x0=s_["x0"])
                o_.solve(default_params(), iters)
                return o_.x.copy(), len(s_["edges"])
            xa, ea = run_on(pix)
            xb, eb = run_on(pix + rng.uniform(-1e-3, 1e-3, pix.shape))
            dd = xa.astype(np.float64) - xb
            rows.append((name, "Delaunay ties of integer-pixel features cut another way (%d features)" % len(pix), float(np.sqrt(np.mean(dd * dd))),
                         float(np.abs(dd).max()), "a frame-level pin needs upstream's triangles; the graph-level pin (this directory) hands the edges over"))
        # ---- triangle stage (row a8): validity flips at the regularised state ----
        Kinv = np.linalg.inv(np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]])).astype(np.float32)
        W, H = g.width, g.height
        tp0 = TriParams(1, 1.57, 0.35, 0.1, 1, 0.333, 1, 0.01, W, H)
        _, v0, _ = triangles(tp0, Kinv, g.pos, base, g.tris)
        for lab, tp in (("oblique_normal_thresh 1.57 -> 1.3 rad", TriParams(1, 1.3, 0.35, 0.1, 1, 0.333, 1, 0.01, W, H)),
                        ("oblique_idepth_diff_factor 0.35 -> 0.2", TriParams(1, 1.57, 0.2, 0.1, 1, 0.333, 1, 0.01, W, H)),
                        ("oblique_idepth_diff_abs 0.1 -> 0.05", TriParams(1, 1.57, 0.35, 0.05, 1, 0.333, 1, 0.01, W, H)),
                        ("edge_length_thresh 0.333 -> 0.1 of the width", TriParams(1, 1.57, 0.35, 0.1, 1, 0.1, 1, 0.01, W, H)),
                        ("min_triangle_idepth 0.01 -> 0.3", TriParams(1, 1.57, 0.35, 0.1, 1, 0.333, 1, 0.3, W, H)),
                        ("oblique filter off", TriParams(0, 1.57, 0.35, 0.1, 1, 0.333, 1, 0.01, W, H))):
            _, v1, _ = triangles(tp, Kinv, g.pos, base, g.tris)
            flips = int((v0 != v1).sum())
            rows.append((name, "triangle filter: " + lab, None, None, "%d of %d validity flags flip (%d valid by default)" % (flips, len(v0), int(v0.sum()))))
    if args.markdown:
        print("# Price list of the recalled choices (oracle, CPU; tools/pin_upstream/price_list.py)\n")
        print("Change of the regularised idepths after the frame's iterations when ONE recalled choice is flipped, against the\n"
              "north_star tolerance 1e-4 RMS.  `> tol` = a disagreement of this kind with upstream would break parity; `< tol` = it\n"
              "would pass unnoticed.  Graphs: BASELINE config 1 (TUM-shaped, 1.2 k vertices, 200 iterations), config 3 (EuRoC-shaped,\n"
              "10 k, 200), config 2 (5 k uniform, 200).\n")
        print("| graph | choice flipped | RMS change of x | max change | vs 1e-4 | note |\n|---|---|---|---|---|---|")
        for name, lab, rms, mx, note in rows:
            if rms is None:
                print("| %s | %s | -- | -- | -- | %s |" % (name, lab, note))
            else:
                print("| %s | %s | %.3g | %.3g | %s | %s |" % (name, lab, rms, mx, "> tol" if rms > TOL else "< tol", note))
    else:
        for name, lab, rms, mx, note in rows:
            if rms is None:
                print("%-6s %-70s %s" % (name, lab, note))
            else:
                print("%-6s %-70s rms %.3g max %.3g %s %s" % (name, lab, rms, mx, ">tol" if rms > TOL else "<tol", note))


if __name__ == "__main__":
    main()
