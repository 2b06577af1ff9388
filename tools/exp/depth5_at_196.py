"""Dev aid: would halo depth 5 pay at the headline's tile size (196 own vertices) if its slot rows fitted 160 KiB?  Graphs of
30-40 k vertices cut into 196-vertex tiles DO fit at depth 5: resident solve at depth 4 vs 5, same tile size."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flame_ros_amd import graphgen  # noqa: E402
from flame_ros_amd.regularizer import GraphRegularizer, default_params  # noqa: E402

p = default_params()
for V in (30000, 36000, 40000):
    g = graphgen.synthetic(V, seed=1)
    for depth in (4, 5, 6):
        try:
            with GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, tile_own=196, tile_depth=depth) as r:
                ms = []
                for _ in range(10):
                    r.step(p, 500)
                    ms.append(r.last_solve_ms()[0])
                print("V %d own 196 depth %d: tiles %d (depth built %d) nt %d ept %d lds %d resident %d: best %.4f median %.4f us/it" % (
                    V, depth, r.info("num_tiles"), r.info("tile_depth"), r.info("tile_threads"), r.info("tile_ept"), r.info("tile_lds_bytes"),
                    r.info("persist_used"), min(ms) * 1e3 / 500, sorted(ms)[5] * 1e3 / 500), flush=True)
        except Exception as e:  # noqa: BLE001
            print("V %d depth %d: %s" % (V, depth, e))
