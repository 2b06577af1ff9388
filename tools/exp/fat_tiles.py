"""Fat resident tiles (r05): graphs beyond 256 x 196 vertices on ONE tile per CU.  Per size: the automatic plan and forced halo
depths, with resident tiles and by launches (persist = 0: the r04 scheme, two rounds of smaller tiles) -- microseconds per PD
iteration, what the plan chose, and every bit against the oracle (sizes in argv, default 65k..220k; FAT_CHECK=0 skips the oracle)."""
import os, sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params
from tests.util import make_oracle, oracle_params
p = default_params()
ITERS = int(os.environ.get("FAT_ITERS", "500"))
CHECK = os.environ.get("FAT_CHECK", "1") != "0"
DEPTHS = [int(d) for d in os.environ.get("FAT_DEPTHS", "0,1,2,3,4").split(",")]


def bench(r, iters, reps=6):
    best = 1e9
    for _ in range(reps):
        r.step(p, iters)
        ms, _l = r.last_solve_ms()
        best = min(best, ms)
    return best * 1e3 / iters


for V in [int(a) for a in (sys.argv[1:] or ["65000", "100000", "130000", "160000", "200000", "220000"])]:
    g = graphgen.synthetic(V, seed=V)
    ref = None
    if CHECK:
        t0 = time.time()
        o = make_oracle(g); o.solve(oracle_params(), ITERS)
        ref = (o.x.copy(), o.w1.copy(), o.w2.copy(), o.q.copy())
        print("V %d: oracle %d iterations in %.1f s" % (V, ITERS, time.time() - t0), flush=True)
    for persist in (0, 1):
        for depth in (DEPTHS if persist else [0]):
            try:
                r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, persist=persist, tile_depth=depth)
            except Exception as e:
                print("   V %d persist %d depth %d: %s" % (V, persist, depth, e)); continue
            bad = ""
            if ref is not None:
                r.step(p, ITERS // 2)          # (a plan's first solve: poll lists in local order, sorted local edges)
                r.step(p, ITERS - ITERS // 2)  # (from the second on: address-sorted poll lists, lane order)
                x, w1, w2, q = r.download()
                same = all(np.array_equal(a.view(np.uint32), b.astype(np.float32).view(np.uint32)) for a, b in zip((x, w1, w2, q), ref))
                bad = "bit-exact" if same else "MISMATCH (x: %d words differ)" % int(np.sum(x.view(np.uint32) != ref[0].astype(np.float32).view(np.uint32)))
            us = bench(r, ITERS)
            print("   V %6d persist %d depth %s -> tiles %4d depth %d cfg %d/%d/%d slot12 %d lds %6d resident %d launches %3d: %7.3f us/it  %s" % (
                V, persist, depth or "auto", r.info("num_tiles"), r.info("tile_depth"), r.info("tile_threads"), r.info("tile_ept"), r.info("tile_vpt"),
                r.info("tile_slot12"), r.info("tile_lds_bytes"), r.info("persist_used"), r.last_solve_ms()[1], us, bad), flush=True)
            r.close()
