"""dev helper: per-tile cycles (in-kernel timeline) against plan features -> cost model for the
balancing pass."""
import os, sys, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flame_ros_amd import graphgen, lib
from flame_ros_amd.regularizer import GraphRegularizer, default_params
g, iters = graphgen.named(sys.argv[1] if len(sys.argv) > 1 else '50k')
for bal in (0, 1):
    r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, profile=1, use_graph=0, balance=bal)
    p = default_params(); r.step(p, iters); d = r.info('tile_depth'); r.step(p, d)
    t = r.plan_array('profile', np.uint64).reshape(-1, 36).astype(np.int64)
    raw = r.plan_array('tiles', np.dtype((np.void, C.sizeof(lib.TileDesc))))
    tiles = (lib.TileDesc * len(raw)).from_buffer_copy(raw.tobytes())
    F = np.array([[x.n_own, x.n_ext, x.e_loc, x.nslots, x.n_upd, x.e_own] for x in tiles], float)
    cyc = (t[:, 35] - t[:, 0]).astype(float)
    load = (t[:, 1] - t[:, 0]).astype(float)
    A = np.c_[F[:, 1], F[:, 2], np.ones(len(F))]
    coef, *_ = np.linalg.lstsq(A, cyc, rcond=None)
    pred = A @ coef
    print('balance', bal, 'cycles p50 %d max %d' % (np.median(cyc), cyc.max()), 'fit cycles ~ %.2f n_ext + %.2f e_loc + %.0f' % tuple(coef),
          'resid std %.0f' % np.std(cyc - pred), 'corr(cost_model, cycles) %.3f' % np.corrcoef(F[:, 2] + 2 * F[:, 1], cyc)[0, 1])
    worst = np.argsort(-cyc)[:5]
    print('   worst tiles', worst.tolist(), 'cycles', cyc[worst].astype(int).tolist(), 'n_ext', F[worst, 1].astype(int).tolist(), 'e_loc', F[worst, 2].astype(int).tolist(), 'load', load[worst].astype(int).tolist())
