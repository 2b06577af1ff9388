import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
from flame_ros_amd.regularizer import GraphRegularizer
h = GraphRegularizer.empty()
rng = np.random.default_rng(0)
for n in (1200, 10000, 50000):
    for kind in ("uniform", "pixels"):
        if kind == "uniform":
            pts = (rng.random((n, 2)) * np.array([640.0, 480.0])).astype(np.float32)
        else:
            pts = np.stack([rng.integers(0, 640, n), rng.integers(0, 480, n)], 1).astype(np.float32)
        print(n, kind, flush=True)
        h.delaunay(pts)
