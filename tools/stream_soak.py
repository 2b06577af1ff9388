"""dev helper (gpurun): a long frame stream of random sizes on ONE handle through the facade-style calls
(graph sync -> solve -> frame_results), single-tile and halo plans mixed; every 25th frame is checked
against the oracle, device memory in use is printed at the start and at the end (leak check)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params, default_sync_params, default_tri_params
from oracle.cbind import SyncParams as OSync, graph_sync as oracle_sync
from oracle import COracle
from tests.util import oracle_params, bits
def kfd_stats():
    """The kernel driver's per-process counters (all KFD processes that own queues: pids here are the host's): milliseconds the
    process' queues were EVICTED, page faults / migrations -- what a whole-launch stall would show up in."""
    import glob
    out = {"evicted_ms": 0, "faults": 0, "page_in": 0, "page_out": 0}
    for d in glob.glob("/sys/class/kfd/kfd/proc/*"):
        if not os.path.isdir(os.path.join(d, "queues")):
            continue
        for key, pat in (("evicted_ms", "stats_*/evicted_ms"), ("faults", "counters_*/faults"), ("page_in", "counters_*/page_in"), ("page_out", "counters_*/page_out")):
            for f in glob.glob(os.path.join(d, pat)):
                try:
                    out[key] += int(open(f).read().strip() or 0)
                except Exception:
                    pass
    return out


n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(11)
# (argv[2] = "facade": the options flame::Flame sets -- small frames on halo tiles, 0.9-1.28 k vertices on persistent tiles)
opts = dict(tile_single_max=640, stream_depth=5) if len(sys.argv) > 2 and sys.argv[2] == "facade" else dict(tile_single_max=2048)
r = GraphRegularizer.empty(device=0, **opts)
Kinv = np.linalg.inv(np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]])).astype(np.float32)
p, sp = default_params(), default_sync_params()
free0 = torch.cuda.mem_get_info()[0]
kfd0 = kfd_stats(); kfd_prev = dict(kfd0)
print("kfd counters at start:", kfd0, flush=True)
t0 = time.perf_counter(); bad = 0
for k in range(n):
    # random sizes, with runs of similar frames in between (partition reuse kicks in on those)
    if k % 7 in (3, 4, 5) and k > 0:
        V = max(200, int(V * rng.uniform(0.95, 1.05)))
    else:
        if len(sys.argv) > 3 and sys.argv[3] == "fat":  # (r05: sizes across the regular / fat tile boundary and the slot layouts)
            V = int(rng.choice([1100, 5000, 30000, 49000, 52000, 70000, 110000, 150000, 190000, 215000], p=[.1] * 10))
        else:
            V = int(rng.choice([300, 900, 1100, 1500, 2600, 5000, 12000, 30000], p=[.15, .15, .15, .15, .15, .1, .1, .05]))
    g = graphgen.synthetic(V, seed=1000 + k)
    var = np.full(g.V, 1e-4, np.float32)
    scale = r.sync_features(g.pos, g.z, var, g.tris, sp)
    try:
        r.step(p, 60, sync=False)
    except Exception as e:  # (which plan was it?)
        print("frame %d V %d E %d: %s | tiles %d depth %d cfg %d/%d/%d lds %d slot12 %d fat %d reused %d on_device %d" % (
            k, g.V, r.E, e, r.info("num_tiles"), r.info("tile_depth"), r.info("tile_threads"), r.info("tile_ept"), r.info("tile_vpt"),
            r.info("tile_lds_bytes"), r.info("tile_slot12"), r.info("tile_fat"), r.info("plan_reused"), r.info("plan_on_device")), flush=True)
        raise
    out = r.frame_results(p, Kinv, default_tri_params(g.width, g.height), scale_back=scale, with_edges=True, with_coverage=True)
    if len(sys.argv) > 3:  # (r05 diagnostics: which frames wait long?)
        wmax = r.info("persist_wait_us_max")
        if wmax > 1000:
            print("frame %4d V %6d: poll wait %d us, solve %.2f ms, tiles %d depth %d cfg %d/%d slot12 %d reused %d timeout %d us" % (
                k, V, wmax, r.last_solve_ms()[0], r.info("num_tiles"), r.info("tile_depth"), r.info("tile_threads"), r.info("tile_ept"),
                r.info("tile_slot12"), r.info("plan_reused"), r.info("persist_timeout_us")), flush=True)
    rec_now = r.info("persist_recovered")
    if rec_now != (rec_seen if k else 0):  # r06: a resident launch gave up and was repeated -- how far had its tiles got?
        print("frame %4d V %6d: REPEATED by launches; solve %.2f ms; tiles %d depth %d cfg %d/%d slot12 %d fat %d lds %d reused %d | give-up %s" % (
            k, V, r.last_solve_ms()[0], r.info("num_tiles"), r.info("tile_depth"), r.info("tile_threads"), r.info("tile_ept"), r.info("tile_slot12"),
            r.info("tile_fat"), r.info("tile_lds_bytes"), r.info("plan_reused"),
            {q: r.info("persist_gave_up_" + q) for q in ("tile", "round", "front_round", "not_started", "rounds", "tiles", "one_xcd", "timeout_us")}), flush=True)
        kn = kfd_stats()
        print("           kfd counters since the last event / start: %s" % {q: kn[q] - kfd_prev[q] for q in kn}, flush=True)
        kfd_prev = kn
    rec_seen = rec_now
    reused = reused + r.info("plan_reused") if k else 0
    persisted = (persisted if k else 0) + r.info("persist_used")
    if k % 25 == 0 or (k % 7 == 4 and k % 3 == 0):
        s = oracle_sync(OSync(0, 0, 1, 0.01), g.pos, g.z, var, g.tris, None)
        o = COracle(g.pos, s["edges"], s["alpha"], s["beta"], s["z"], s["wgt"], x0=s["x0"]); o.solve(oracle_params(), 60)
        nb = int((bits(out[2]) != bits(o.x)).sum()); bad += nb
        print("frame %4d V %6d  plan_on_device %d reused %d  bad words %d" % (k, V, r.info("plan_on_device"), r.info("plan_reused"), nb), flush=True)
dt = time.perf_counter() - t0
free1 = torch.cuda.mem_get_info()[0]
kn = kfd_stats()
print("kfd counters over the run:", {q: kn[q] - kfd0[q] for q in kn}, flush=True)
print("torn-read debug build: %d; torn hand-off entries counted: %d; longest poll wait %d us (time-out %d us)" % (
    r.info("torn_check_build"), r.info("persist_torn"), r.info("persist_wait_us_max"), r.info("persist_timeout_us")))
print("frames %d in %.1f s (incl. graph generation); %d plans from a reused partition; device memory in use changed by %.1f MiB; bad words %d; %d solves on persistent tiles, %d of them repeated" % (n, dt, reused, (free0 - free1) / 2**20, bad, persisted, r.info("persist_recovered")))
r.close()
