"""dev helper (gpurun): stress of the device graph sync + plan builder (single-launch scans, partition
reuse, speculative sizes, predicted edge count) -- N frames of 8 k..60 k vertices on ONE handle, EVERY
frame's edges and an 8-iteration solve checked against the oracle."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params, default_sync_params
from oracle.cbind import SyncParams as OSync, graph_sync as oracle_sync
from oracle import COracle
from tests.util import oracle_params, bits
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(5)
r = GraphRegularizer.empty(device=0)
p, sp = default_params(), default_sync_params()
bad = reused = 0
V = 20000
t0 = time.perf_counter()
for k in range(n):
    if k % 5 == 0:
        V = int(rng.choice([8000, 14000, 20000, 33000, 50000, 60000]))
    else:
        V = int(V * rng.uniform(0.97, 1.03))
    g = graphgen.synthetic(V, seed=3000 + k)
    tris = g.tris if k % 11 else g.tris[rng.random(len(g.tris)) > 0.1]   # sometimes a mesh with holes
    var = np.full(g.V, 1e-4, np.float32)
    r.sync_features(g.pos, g.z, var, tris, sp)
    r.step(p, 8, sync=False)
    x = r.download(with_q=False)[0]
    s = oracle_sync(OSync(0, 0, 1, 0.01), g.pos, g.z, var, tris, None)
    o = COracle(g.pos, s["edges"], s["alpha"], s["beta"], s["z"], s["wgt"], x0=s["x0"]); o.solve(oracle_params(), 8)
    nb = int((bits(x) != bits(o.x)).sum()) + int(not np.array_equal(r.edges(), s["edges"]))
    bad += nb
    reused += r.info("plan_reused")
    if nb or k % 20 == 0:
        print("frame %4d V %6d tiles %4d reused %d bad %d" % (k, V, r.info("num_tiles"), r.info("plan_reused"), nb), flush=True)
print("frames %d, %d reused partitions, bad %d, %.1f s" % (n, reused, bad, time.perf_counter() - t0))
sys.exit(1 if bad else 0)
