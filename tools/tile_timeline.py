"""dev helper: in-kernel timeline of the tile kernel (cycles) per tile."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params
import argparse
ap = argparse.ArgumentParser(); ap.add_argument('--workload', default='50k'); ap.add_argument('--own', type=int, default=0); ap.add_argument('--depth', type=int, default=0); ap.add_argument('--nt', type=int, default=0); ap.add_argument('--order', type=int, default=-1)
a = ap.parse_args()
g, iters = graphgen.named(a.workload)
opts = dict(profile=1, use_graph=0)
if a.own: opts['tile_own'] = a.own
if a.depth: opts['tile_depth'] = a.depth
if a.nt: opts['tile_threads'] = a.nt
if a.order >= 0: opts['order_mode'] = a.order
r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, **opts)
p = default_params()
r.step(p, iters); r.step(p, iters)
d = r.info('tile_depth')
r.step(p, d)
t = r.plan_array('profile', np.uint64).reshape(-1, 36).astype(np.int64)
ms, n = r.last_solve_ms()
print('tiles', len(t), 'depth', d, 'nt', r.info('tile_threads'), 'ept', r.info('tile_ept'), 'vpt', r.info('tile_vpt'), 'launch ms', ms)
t0 = t[:, 0].min()
load = t[:, 1] - t[:, 0]
store = t[:, 35] - t[:, 2 * d + 1]
print('start skew (cycles) p50 %d max %d' % (np.median(t[:, 0] - t0), (t[:, 0] - t0).max()))
print('load   p50 %d  max %d' % (np.median(load), load.max()))
for k in range(1, d + 1):
    pd = t[:, 2 * k] - t[:, 2 * k - 1]; pp = t[:, 2 * k + 1] - t[:, 2 * k]
    print('iter%d  D p50 %d max %d | P p50 %d max %d' % (k, np.median(pd), pd.max(), np.median(pp), pp.max()))
print('store  p50 %d  max %d' % (np.median(store), store.max()))
print('total  p50 %d  max %d ; span %d' % (np.median(t[:, 35] - t[:, 0]), (t[:, 35] - t[:, 0]).max(), t[:, 35].max() - t0))
# slowest tiles against the median tile: what they are made of (TileDesc = 13 header ints + rings/levels)
td = r.plan_array('tiles', np.int32).reshape(len(t), -1)
tot = t[:, 35] - t[:, 0]
dsum = sum(t[:, 2 * k] - t[:, 2 * k - 1] for k in range(1, d + 1))
psum = sum(t[:, 2 * k + 1] - t[:, 2 * k] for k in range(1, d + 1))
order = np.argsort(tot)
print('tile   n_own n_ext n_upd e_own e_loc nslots |  load     D     P store total')
for i in list(order[-6:][::-1]) + [order[len(order) // 2], order[0]]:
    print('%5d %6d %5d %5d %5d %5d %6d | %5d %5d %5d %5d %5d' % (i, td[i, 1], td[i, 2], td[i, 6], td[i, 4], td[i, 5], td[i, 12],
          load[i], dsum[i], psum[i], store[i], tot[i]))
for name, col in (('n_ext', td[:, 2]), ('e_loc', td[:, 5]), ('nslots', td[:, 12])):
    print('corr(total, %s) = %.2f   corr(load, %s) = %.2f' % (name, np.corrcoef(tot, col)[0, 1], name, np.corrcoef(load, col)[0, 1]))
