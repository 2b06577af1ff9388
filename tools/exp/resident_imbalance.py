"""Per-launch time of the tile kernel on a 50 k / 10 k graph by how its partition was balanced: the upload
path's passes (unweighted, weighted, refined), none, and a frame stream's (one pass from the cost grid, then
reused).  Same timing for all: replayed solves of a resident plan, lane order off."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params, default_sync_params
import torch
p, sp = default_params(), default_sync_params()


def timed(r, it):
    r.step(p, it); r.step(p, it); r.step(p, it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8): r.step(p, it)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    return "imbalance %d%% ept %d tiles %d depth %d: %.3f ms per solve, %.2f us/launch" % (
        r.info("tile_imbalance_pct"), r.info("tile_ept"), r.info("num_tiles"), r.info("tile_depth"), dt * 1e3,
        dt * 1e6 / -(-it // r.info("tile_depth")))


for name in ("50k", "euroc"):
    g, it = graphgen.named(name)
    for kw in (dict(), dict(balance=0)):
        r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, lane_order=0, **kw)
        print(name, "upload", kw, timed(r, it))
        r.close()
    r = GraphRegularizer.empty(device=0, lane_order=0)
    for k in range(6):
        f = graphgen.named(name, seed=10 + k)[0]
        r.sync_features(f.pos, f.z, np.full(f.V, 1e-4, np.float32), f.tris, sp)
        if k in (0, 1, 5):
            print(name, "stream frame %d reused %d" % (k, r.info("plan_reused")), timed(r, it))
    r.close()
