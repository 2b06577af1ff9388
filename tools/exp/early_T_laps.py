"""r06 experiment aid (tools/exp/early_T.patch applied): one keep-mode frame stream with / without the early T, plan_timing on at the end."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from flame_ros_amd.regularizer import GraphRegularizer, default_params, default_sync_params
sp, p = default_sync_params(), default_params()
n = int(sys.argv[1]); early = int(sys.argv[2])
rng = np.random.default_rng(n)
pts = (rng.random((n, 2)) * np.array([640.0, 480.0])).astype(np.float32)
mu = (0.5 + 0.001 * pts[:, 0] + 0.05 * rng.standard_normal(n)).astype(np.float32)
var = np.full(n, 1e-4, np.float32)
with GraphRegularizer.empty() as h:
    h.set_option("delaunay_early_T", early)
    for rep in range(12):
        if rep == 10 and len(sys.argv) < 4: h.set_option("plan_timing", 3)
        T = h.delaunay_keep(pts)
        h.sync_features(pts, mu, var, T, sp)
        h.step(p, 50, sync=False)
        h.delaunay_list(); h.download()
