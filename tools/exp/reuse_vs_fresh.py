"""What a reused partition costs the SOLVE on i.i.d. frames: sync and solve+results time of a 50 k stream with
plan_reuse 1 (the partition of the previous frame) and 0 (bisected and balanced anew every frame)."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params, default_sync_params, default_tri_params
name = sys.argv[1] if len(sys.argv) > 1 else "50k"
frames = [graphgen.named(name, seed=k) for k in range(4)]
p, sp = default_params(), default_sync_params()
Kinv = np.linalg.inv(np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]])).astype(np.float32)
for reuse in (1, 0, 1, 0):
    r = GraphRegularizer.empty(device=0, tile_single_max=896, stream_depth=5, plan_reuse=reuse)
    ts, tv, imb, ept = [], [], [], []
    for k in range(24):
        g = frames[k & 3][0]
        tp = default_tri_params(g.width, g.height)
        var = np.full(g.V, 1e-4, np.float32)
        t0 = time.perf_counter()
        scale = r.sync_features(g.pos, g.z, var, g.tris, sp)
        t1 = time.perf_counter()
        r.step(p, frames[0][1], sync=False)
        out = r.frame_results(p, Kinv, tp, scale_back=scale, with_edges=True, with_coverage=True)
        t2 = time.perf_counter()
        if k >= 4:
            ts.append((t1 - t0) * 1e3); tv.append((t2 - t1) * 1e3); imb.append(r.info("tile_imbalance_pct")); ept.append(r.info("tile_ept"))
    print("plan_reuse %d: sync p50 %.3f  solve+results p50 %.3f  total p50 %.3f  imbalance %s ept %s" % (
        reuse, np.median(ts), np.median(tv), np.median(np.array(ts) + np.array(tv)), sorted(set(imb)), sorted(set(ept))))
    r.close()
