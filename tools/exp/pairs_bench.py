"""EXPERIMENT (r05): two resident workgroups per CU (FLAME_HIP_PERSIST_PAIRS=1): half-size tiles, 512 threads, one hides the
other's hand-off.  us per iteration at 50 k / 10 k / 5 k against the default plan; bits against the oracle at 50 k."""
import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params
from tests.util import make_oracle, oracle_params
p = default_params()

def bench(r, iters, reps=8):
    best = 1e9
    for _ in range(reps):
        r.step(p, iters); best = min(best, r.last_solve_ms()[0])
    return best * 1e3 / iters

for name, cfgs in (("50k", [dict(), dict(tile_own=98, tile_depth=3, tile_threads=512), dict(tile_own=98, tile_depth=2, tile_threads=512), dict(tile_own=110, tile_depth=3, tile_threads=512), dict(tile_own=128, tile_depth=3, tile_threads=512)]),
                   ("euroc", [dict(), dict(tile_own=20, tile_depth=4, tile_threads=256), dict(tile_own=20, tile_depth=5, tile_threads=256), dict(tile_own=24, tile_depth=4, tile_threads=512)]),
                   ("5k", [dict(), dict(tile_own=10, tile_depth=4, tile_threads=256), dict(tile_own=16, tile_depth=4, tile_threads=256), dict(tile_own=16, tile_depth=5, tile_threads=256)])):
    g, it = graphgen.named(name)
    o = make_oracle(g); o.solve(oracle_params(), 2 * it)
    for kw in cfgs:
        try:
            r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, **kw)
        except Exception as e:
            print(name, kw, "ERR", e); continue
        r.step(p, it); r.step(p, it)
        x = r.download()[0]
        same = np.array_equal(x.view(np.uint32), o.x.astype(np.float32).view(np.uint32))
        us = bench(r, it)
        print("%-5s %-55s tiles %3d depth %d cfg %d/%d lds %6d resident %d launches %3d: %.3f us/it %s wait_max %d us gave_up %d" % (
            name, kw, r.info("num_tiles"), r.info("tile_depth"), r.info("tile_threads"), r.info("tile_ept"), r.info("tile_lds_bytes"),
            r.info("persist_used"), r.last_solve_ms()[1], us, "bit-exact" if same else "MISMATCH", r.info("persist_wait_us_max"), r.info("persist_gave_up")), flush=True)
        r.close()
