"""Dev aid: microseconds per PD iteration of the resident tiles on the BASELINE graphs, with the round split of a few
tiles and the bits against the oracle -- one line per graph; run once per library variant (FLAME_HIP_LIB).
  python tools/exp/resident_ab.py [names...] [--opt k=v ...] [--reps N]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flame_ros_amd import graphgen  # noqa: E402
from flame_ros_amd.regularizer import GraphRegularizer, default_params  # noqa: E402
from tests.util import make_oracle, oracle_params  # noqa: E402

args, names, kw, reps = sys.argv[1:], [], {}, 12
i = 0
while i < len(args):
    if args[i] == "--opt":
        k, v = args[i + 1].split("=")
        kw[k] = int(v)
        i += 2
    elif args[i] == "--reps":
        reps = int(args[i + 1])
        i += 2
    else:
        names.append(args[i])
        i += 1
p = default_params()
tag = os.path.basename(os.environ.get("FLAME_HIP_LIB", "default"))
for name in (names or ["50k", "euroc", "5k", "tum"]):
    g, it = graphgen.named(name)
    o = make_oracle(g)
    o.solve(oracle_params(), it)
    with GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, **kw) as r:
        r.step(p, it)
        x, w1, w2, q = r.download()
        ok = np.array_equal(x.view(np.uint32), o.x.view(np.uint32)) and np.array_equal(q.view(np.uint32), o.q.view(np.uint32))
        ms = []
        for _ in range(reps):
            r.step(p, it)
            ms.append(r.last_solve_ms()[0])
        ms.sort()
        nt, d = r.info("num_tiles"), r.info("tile_depth")
        rows = []
        if r.info("persist_used"):
            for t in sorted(set(int(v) for v in np.linspace(0, nt - 1, 8))):
                r.set_option("persist_prof", t + 1)
                r.step(p, it)
                v = [r.info("persist_prof_%d" % k) for k in range(5)]
                if v[3] > 1:
                    n = v[3] - 1
                    rows.append([v[0] / n / 100.0, v[1] / n / 100.0, v[2] / n / 100.0, v[4] / n])
            r.set_option("persist_prof", 0)
        sp = np.median(np.asarray(rows), axis=0) if rows else [float("nan")] * 4
        print("%-12s %-6s V %6d tiles %3d depth %d nt %4d ept %d resident %d: best %.4f median %.4f us/it | round: iterate %.2f poll %.2f apply %.2f us, passes %.2f | bit-exact %s recovered %d" % (
            tag, name, g.V, nt, d, r.info("tile_threads"), r.info("tile_ept"), r.info("persist_used"), ms[0] * 1e3 / it,
            ms[len(ms) // 2] * 1e3 / it, sp[0], sp[1], sp[2], sp[3], ok, r.info("persist_recovered")), flush=True)
