"""r06: keep-mode flame_hip_delaunay with T handed out before the stars (option delaunay_early_T = 1) against after them (0):
host milliseconds of delaunay_keep / sync_features / 50 iterations + download, frame after frame on one handle.
(Needs a library built with tools/exp/early_T.patch applied: the product has no such option.)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flame_ros_amd.regularizer import GraphRegularizer, default_params, default_sync_params  # noqa: E402

sp, p = default_sync_params(), default_params()
for n in (1200, 10000, 50000):
    rng = np.random.default_rng(n)
    frames = []
    for f in range(8):
        pts = (rng.random((n, 2)) * np.array([640.0, 480.0])).astype(np.float32)
        mu = (0.5 + 0.001 * pts[:, 0] + 0.05 * rng.standard_normal(n)).astype(np.float32)
        frames.append((pts, mu, np.full(n, 1e-4, np.float32)))
    for early in (0, 1, 0, 1):
        with GraphRegularizer.empty() as h:
            h.set_option("delaunay_early_T", early)
            rows = []
            for rep in range(40):
                pts, mu, var = frames[rep % len(frames)]
                t0 = time.perf_counter()
                T = h.delaunay_keep(pts)
                t1 = time.perf_counter()
                h.sync_features(pts, mu, var, T, sp)
                t2 = time.perf_counter()
                h.step(p, 50, sync=False)
                tris = h.delaunay_list()
                x = h.download()[0]
                t3 = time.perf_counter()
                rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3))
            r = np.median(np.asarray(rows[8:]), axis=0)
            print("V %6d early %d: delaunay %.3f  sync %.3f  solve+list+download %.3f  total %.3f ms  (early used %d, mismatch %d)" % (
                n, early, r[0], r[1], r[2], r[3], h.info("delaunay_early_used"), h.info("delaunay_early_mismatch")), flush=True)
