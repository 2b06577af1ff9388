"""flame_hip_delaunay: host time per call (copies included) at the BASELINE sizes, uniform features and integer pixels."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flame_ros_amd.regularizer import GraphRegularizer

h = GraphRegularizer.empty()
rng = np.random.default_rng(0)
for n in (1200, 5000, 10000, 50000, 200000):
    for kind in ("uniform", "pixels"):
        if kind == "uniform":
            pts = (rng.random((n, 2)) * np.array([640.0, 480.0])).astype(np.float32)
        else:
            pts = np.stack([rng.integers(0, 640, n), rng.integers(0, 480, n)], 1).astype(np.float32)
        for _ in range(3):
            t = h.delaunay(pts)
        ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            t = h.delaunay(pts)
            ts.append((time.perf_counter() - t0) * 1e3)
        print("%7d %-8s T %7d hull %5d  p50 %.3f ms  min %.3f ms (library: %d us)" % (n, kind, len(t), h.info("delaunay_hull"), np.median(ts), min(ts), h.info("delaunay_us")), flush=True)
# skewed inputs: the grid assumes nothing, but a frame whose features sit in a few cells degrades to scanning those cells
for name, pts in (("3000 in a 3-px blob + 2500 spread", np.concatenate([rng.normal((300, 300), 3, (3000, 2)), rng.random((2500, 2)) * np.array([640.0, 480.0])])),
                  ("10000 in a 1-px blob + 4 corners", np.concatenate([rng.normal((320, 240), 0.3, (10000, 2)), np.array([[0, 0], [639, 0], [0, 479], [639, 479.0]])])),
                  ("10000 on 8 image rows", np.stack([rng.random(10000) * 640, rng.integers(0, 8, 10000) * 60.0], 1))):
    pts = pts.astype(np.float32)
    for _ in range(2):
        t = h.delaunay(pts)
    t0 = time.perf_counter(); t = h.delaunay(pts); dt = (time.perf_counter() - t0) * 1e3
    print("%-36s V %6d T %7d  %.3f ms" % (name, len(pts), len(t), dt), flush=True)
