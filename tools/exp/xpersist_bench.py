"""Resident graphs of 33..256 tiles: launches per round against ONE launch of tiles resident over ALL XCDs (option
persist = 3: L2 hand-offs inside an XCD's eighth of the tiles, uncached mirrors across XCD borders).  us per iteration,
the bits against the oracle after the first solve and against the launches after every later one.
  python tools/exp/xpersist_bench.py [names...] [--depth D] [--own N] [--opt key=value ...]"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params
from tests.util import make_oracle, oracle_params
args = sys.argv[1:]
kw = {}
names = []
pv = 1
i = 0
while i < len(args):
    if args[i] == "--depth": kw["tile_depth"] = int(args[i + 1]); i += 2
    elif args[i] == "--own": kw["tile_own"] = int(args[i + 1]); i += 2
    elif args[i] == "--threads": kw["tile_threads"] = int(args[i + 1]); i += 2
    elif args[i] == "--persist": pv = int(args[i + 1]); i += 2
    elif args[i] == "--opt": k_, v_ = args[i + 1].split("="); kw[k_] = int(v_); i += 2  # any handle option (poll_delay=3, persist_prof=<tile+1> ...)
    else: names.append(args[i]); i += 1
p = default_params()
for name in (names or ["5k", "euroc", "50k"]):
    g, it = graphgen.named(name)
    o = make_oracle(g); o.solve(oracle_params(), it)
    ref = None
    for persist in (0, pv):
        r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, persist=persist, **kw)
        r.step(p, it)
        x, w1, w2, q = r.download()
        ok = np.array_equal(x.view(np.uint32), o.x.view(np.uint32)) and np.array_equal(q.view(np.uint32), o.q.view(np.uint32))
        best = 1e9
        for _ in range(8):
            r.step(p, it)
            best = min(best, r.last_solve_ms()[0])
        x2, _, _, q2 = r.download()
        if ref is None: ref = (x2, q2); ok2 = True
        else: ok2 = np.array_equal(x2.view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(q2.view(np.uint32), ref[1].view(np.uint32))
        prof = ""
        if persist and kw.get("persist_prof"):
            t = [r.info("persist_prof_%d" % k) for k in range(4)]
            n = max(t[3] - 1, 1)
            prof = "  tile %s rounds %d: iterate+store %.2f us, poll %.2f, apply+barrier %.2f per round" % (
                kw["persist_prof"] - 1, t[3], t[0] / n / 100., t[1] / n / 100., t[2] / n / 100.)
        print("%-6s persist %d: %.3f us/it (%.3f ms per %d)  tiles %d depth %d threads %d used %d recovered %d  bit-exact: first solve vs oracle %s, 9th vs launches %s%s" % (
            name, persist, best * 1e3 / it, best, it, r.info("num_tiles"), r.info("tile_depth"), r.info("tile_threads"), r.info("persist_used"),
            r.info("persist_recovered"), ok, ok2, prof), flush=True)
        r.close()
