import numpy as np, torch, sys
sys.path.insert(0, '.')
from flame_ros_amd import dist as fdist
from flame_ros_amd.regularizer import GraphRegularizer, default_params
from oracle import COracle
from oracle.cbind import default_params as op
from tests.util import graphgen
world, depth = 4, 3
g = graphgen.synthetic(6000, seed=21)
part = fdist.rcb_parts(g.pos, world)
for r in range(world):
    s = fdist.build_subdomain(g.pos, g.edges, part, r, depth)
    args = (g.pos[s.vid], s.edges, g.alpha[s.eid], g.beta[s.eid], g.z[s.vid], g.wgt[s.vid])
    deg = np.bincount(s.edges.ravel(), minlength=len(s.vid))
    print('sub', r, 'V', len(s.vid), 'E', len(s.eid), 'deg0', int((deg == 0).sum()), 'maxdeg', deg.max())
    for opts in (dict(), dict(path=1), dict(use_graph=0), dict(tile_own=64, tile_depth=4)):
        for n in (3, 1, 4):
            o = COracle(*args); o.solve(op(), n)
            R = GraphRegularizer(*args, **opts); R.step(default_params(), n)
            x, w1, w2, q = R.download()
            bad = int((x.view(np.uint32) != o.x.view(np.uint32)).sum())
            badq = int((q.view(np.uint32) != o.q.view(np.uint32)).sum())
            print('   ', opts, 'n', n, 'tiles', R.info('num_tiles'), 'depth', R.info('tile_depth'), 'nt', R.info('tile_threads'), 'bad x', bad, 'bad q', badq)
            if bad:
                idx = np.flatnonzero(x.view(np.uint32) != o.x.view(np.uint32))[:8]
                print('      idx', idx, 'ring', s.ring[idx], 'deg', deg[idx])
