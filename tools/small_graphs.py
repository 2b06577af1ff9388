import sys, time, numpy as np
sys.path.insert(0, '.')
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params
p = default_params()
for name, g in (("tum1200", graphgen.dataset_shaped(640, 480, 16)), ("euroc5640", graphgen.dataset_shaped(752, 480, 8)), ("v2000", graphgen.synthetic(2000))):
    for opts in ({}, dict(tile_own=32, tile_depth=4), dict(tile_own=48, tile_depth=4), dict(tile_own=24, tile_depth=3), dict(tile_own=64, tile_depth=5)):
        r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, **opts)
        for _ in range(3): r.step(p, 200)
        ts = []
        for _ in range(10):
            r.step(p, 200); ts.append(r.last_solve_ms()[0])
        print(name, g.V, opts, 'tiles', r.info('num_tiles'), 'depth', r.info('tile_depth'), 'ms/200it %.3f' % np.median(ts), 'us/iter %.3f' % (np.median(ts) * 5))
