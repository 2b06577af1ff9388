#!/usr/bin/env python
"""bench.py -- primal-dual iterations/s of the NLTGV2-L1 graph regulariser on MI355X.

One "step" = one pass of the hot path over one frame: `iters` PD iterations on a resident
synthetic Delaunay graph (BASELINE.json config: 50k vertices / 150k edges / 500 iterations; the
cost of an iteration is data independent, so successive steps simply keep iterating the resident
state).  N>1 GPUs: one process per GPU, every rank regularises its own frame (replicas, weak
scaling, no data-path collective).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(workload, batch_win, iters, budget_s=12.0):
    """The oracle (our restatement of upstream's sequential step(); kind = "port") timed on the
    host cores of this box, one thread (the contract's figure) plus the OpenMP variant at
    {1,4,8,16,32,64} threads.  Runs in its own process (oracle/cpu_baseline.py): pinned OpenMP
    threads (OMP_PROC_BIND/OMP_PLACES must be set before libgomp starts, and torch has already
    loaded one here) and a -march=native build for this host."""
    import subprocess
    env = dict(os.environ, OMP_PROC_BIND="close", OMP_PLACES="cores")
    env.pop("OMP_NUM_THREADS", None)
    cmd = [sys.executable, "-m", "oracle.cpu_baseline", "--budget", str(budget_s), "--iters", str(iters)]
    cmd += ["--batch-win", str(batch_win)] if batch_win else ["--workload", workload]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        raise RuntimeError("cpu baseline leg failed: " + out.stderr[-2000:])
    return json.loads(out.stdout.strip().splitlines()[-1])


def parity_check(g, iters, device, opts):
    """Accuracy line of SURVEY.md 8(d): the GPU result of one frame (fresh state, `iters`
    iterations) against the oracle on the same inputs -- the oracle as checker, not as product."""
    import numpy as np
    from flame_ros_amd.regularizer import GraphRegularizer, default_params
    from oracle import COracle
    from oracle.cbind import default_params as oparams
    o = COracle(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt)
    o.solve(oparams(), iters)
    with GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, device=device, **opts) as r:
        r.step(default_params(), iters)
        x = r.download(with_q=False)[0]
        smooth, data = r.costs(default_params())
    d = x.astype(np.float64) - o.x
    so, do = o.costs(oparams())
    return {"rms_idepth": float(np.sqrt(np.mean(d * d))), "max_abs": float(np.abs(d).max()),
            "bit_exact": bool(np.array_equal(x.view(np.uint32), o.x.view(np.uint32))),
            "tolerance_rms": 1e-4, "iterations": iters,
            "smoothness_cost": smooth, "data_cost": data,
            "oracle_smoothness_cost": so, "oracle_data_cost": do}


def _summarize():
    """profiles/summarize.py as a module: the LDS floor / roofline functions live beside the counter summaries they are
    checked against (`python profiles/summarize.py --roofline <summary.json>` recomputes a line's block)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("flame_profiles_summarize", os.path.join(ROOT, "profiles", "summarize.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def kernel_src_sha():
    """Hash of the sources the committed PMC passes belong to (profiles/summarize.py writes the same):
    counters of an older kernel are not quoted."""
    return _summarize().kernel_src_sha()


def lds_block(r, us_per_iteration, iters_per_launch, counters=None, iterate_frac=None, measured_hbm_frac=None,
              contract_frac=None, valu_counters=None):
    """The on-chip roofline of a tile-path line: the LDS issue floor of the vertices / edges the
    tiles OWN (from the plan, priced with the guide's LDS table, slowest CU) over the measured shader cycles per
    iteration -- a fraction that cannot exceed 1 -- with the counters' redundancy / busy / conflict shares beside it."""
    import numpy as np
    S = _summarize()
    fl = S.lds_floor(r.plan_array("tiles", np.int32), r.plan_array("t_srow", np.uint32), num_cus=r.info("num_cus") or 256)
    if not fl:
        return None
    blk = S.lds_roofline(fl, us_per_iteration, r.info("clock_khz") / 1e3, iters_per_launch, counters, iterate_frac,
                         measured_hbm_frac, contract_frac, valu_counters)
    blk["lds_floor"] = fl
    return blk


def profiled_counters(workload, kernel):
    """HBM bytes per launch of `kernel` (separate FETCH_SIZE / WRITE_SIZE passes, FETCH x2 gfx950
    correction) and its LDS counters per launch, from the newest committed rocprofv3 PMC summary of
    the same workload (profiles/collect.sh).  PMC counters cannot be read inside an un-profiled run;
    the summary is only used when it was taken from THESE sources (kernel_src_sha), else None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_summary.json" % workload))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        rec = {"source": os.path.relpath(f, ROOT), "kernel_src_sha": d.get("kernel_src_sha"), "bytes_per_launch": None,
               "lds": None, "valu": None}
        for k, v in d.get("traffic_bytes_per_launch", {}).items():
            if kernel in k:
                rec["bytes_per_launch"] = v
        for k, v in d.get("lds_per_launch", {}).items():
            if kernel in k:
                rec["lds"] = v
        for k, v in d.get("valu_per_launch", {}).items():
            if kernel in k:
                rec["valu"] = v
        best = rec
    if best and best["kernel_src_sha"] != kernel_src_sha():
        return {"source": best["source"], "stale": True, "bytes_per_launch": None, "lds": None}
    return best


LDS_PMC = ("SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES")
VALU_PMC = ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES")


def live_hbm_traffic(kernel, child_args, budget_s=110.0):
    """Counters of `kernel` measured IN THIS RUN: short child runs of this script under `rocprofv3 --kernel-trace --pmc ...`, one
    pass per counter set as profiles/collect.sh (FETCH_SIZE and WRITE_SIZE separately, as MI355X_MICROARCH.md prescribes --
    FETCH x 2 is the guide's gfx950 correction, both in KiB; then the LDS and the VALU / wait sets), per launch.  Bounded in time
    and never fatal: None when rocprofv3 is missing or the HBM passes fail -- the caller then quotes the committed profile of
    the same sources (profiled_counters); the LDS / VALU sets are optional extras of the same kind ("lds" / "valu" = None)."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    if os.environ.get("FLAME_BENCH_CHILD") or not shutil.which("rocprofv3"):
        return None
    if os.environ.get("ROCP_TOOL_LIBRARIES") or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None  # (this run is itself being profiled: no profiler inside a profiler)
    t_end = time.perf_counter() + budget_s

    def one_pass(names):
        left = t_end - time.perf_counter()
        if left < 20.0:
            return None
        d = tempfile.mkdtemp(prefix="flame_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + list(names) + ["--output-format", "csv", "-d", d, "-o", "p", "--",
               sys.executable, os.path.abspath(__file__), "--no-cpu", "--no-facade", "--steps", "4", "--warmup", "1"] + child_args
        try:
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                                    "MASTER_ADDR", "MASTER_PORT", "FLAME_BENCH_FORCE_PARTITION")}
            pr = subprocess.Popen(cmd, cwd=ROOT, env=dict(env, FLAME_BENCH_CHILD="1", TMPDIR="/tmp"),
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                pr.wait(timeout=min(left - 5.0, 75.0))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(pr.pid, signal.SIGKILL)  # (the group this call started: rocprofv3 and its child)
                except OSError:
                    pass
                pr.wait(timeout=10)
                return None
            acc, rows, disp = {}, 0, set()
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get("Counter_Name") in names and kernel in r.get("Kernel_Name", ""):
                        acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                        rows += 1
                        disp.add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
            if not rows or not disp:
                return None
            return {k: v / len(disp) for k, v in acc.items()}, len(disp)
        except Exception:  # noqa: BLE001 -- a side measurement: the committed profile is quoted instead
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)

    fe, wr = one_pass(("FETCH_SIZE",)), None
    if fe:
        wr = one_pass(("WRITE_SIZE",))
    if not fe or not wr:
        return None
    out = {"bytes_per_launch": (2.0 * fe[0]["FETCH_SIZE"] + wr[0]["WRITE_SIZE"]) * 1024.0, "lds": None, "valu": None,
           "source": "live: rocprofv3 --pmc passes of this run (FETCH_SIZE, WRITE_SIZE separately, %d launches each; FETCH x 2, KiB)" % fe[1]}
    lds = one_pass(LDS_PMC)
    if lds and lds[0].get("SQ_INSTS_LDS", 0) > 0:
        out["lds"] = dict(lds[0], launches=lds[1])
    valu = one_pass(VALU_PMC)
    if valu and valu[0].get("SQ_INSTS_VALU", 0) > 0:
        out["valu"] = dict(valu[0], launches=valu[1])
    return out


def tile_phase_split(g, iters, device, opts, launch_us):
    """In-kernel timeline of ONE tile launch (s_memtime stamps written by the kernel when the handle
    has profile=1; tools/tile_timeline.py prints the long form): p50 over tiles of the load phase,
    the iterations (phase D + phase P), the store phase, in shader-clock cycles; what is left of the
    measured launch period is the kernel boundary (launch gap + end-of-kernel release)."""
    import numpy as np
    from flame_ros_amd.regularizer import GraphRegularizer, default_params
    o = dict(opts, profile=1, use_graph=0)
    with GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, device=device, **o) as r:
        pr = default_params()
        r.step(pr, iters)
        r.step(pr, iters)  # (lane order applied from the second solve on)
        d = r.info("tile_depth")
        if d <= 0:
            return None
        r.step(pr, d)
        t = r.plan_array("profile", np.uint64).reshape(-1, 36).astype(np.int64)
        mhz = r.info("clock_khz") / 1e3
    t = t[t[:, 35] > 0]
    if not len(t):
        return None
    load = t[:, 1] - t[:, 0]
    it = t[:, 2 * d + 1] - t[:, 1]
    store = t[:, 35] - t[:, 2 * d + 1]
    tot = t[:, 35] - t[:, 0]
    med = lambda a: float(np.median(a))  # noqa: E731
    in_us = med(tot) / mhz
    return {"unit": "shader-clock cycles, p50 over tiles of one launch", "iterations_in_launch": int(d),
            "load": med(load), "iterate": med(it), "store": med(store), "in_kernel": med(tot),
            "in_kernel_slowest_tile": float(tot.max()),
            "clock_mhz_assumed": mhz, "in_kernel_us_at_assumed_clock": in_us,
            "boundary_us": max(0.0, launch_us - float(tot.max()) / mhz),
            "load_frac": med(load) / med(tot), "iterate_frac": med(it) / med(tot), "store_frac": med(store) / med(tot),
            "iterate_frac_of_launch": (med(it) / mhz) / launch_us,
            "note": "boundary_us = measured launch period - slowest tile's in-kernel time at the assumed clock"}


def resident_round_split(g, iters, device, opts, solve_us):
    """Where a round of the RESIDENT tiles goes (option persist_prof: one tile per solve stamps wall_clock64 around its
    rounds): p50 over a sample of tiles of {iterations + stores, poll of the halo entries until they carry the round's
    tag, halo applied + workgroup barrier}, microseconds per round.  iterate_frac = the first part's share."""
    import numpy as np
    from flame_ros_amd.regularizer import GraphRegularizer, default_params
    with GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, device=device, **opts) as r:
        pr = default_params()
        r.step(pr, iters)
        r.step(pr, iters)  # (lane order applied from the second solve on)
        nt, d = r.info("num_tiles"), r.info("tile_depth")
        if not r.info("persist_used") or d <= 0:
            return None
        rows = []
        for t in sorted(set(int(x) for x in np.linspace(0, nt - 1, 12))):
            r.set_option("persist_prof", t + 1)
            r.step(pr, iters)
            v = [r.info("persist_prof_%d" % k) for k in range(5)]
            if v[3] > 1:
                rows.append([v[0] / (v[3] - 1) / 100.0, v[1] / (v[3] - 1) / 100.0, v[2] / (v[3] - 1) / 100.0, v[4] / (v[3] - 1)])
        r.set_option("persist_prof", 0)
    if not rows:
        return None
    a = np.median(np.asarray(rows), axis=0)
    rounds = -(-iters // d)
    return {"unit": "us per round, p50 over %d sampled tiles (wall_clock64 stamps, 10 ns ticks)" % len(rows),
            "iterations_per_round": int(d), "rounds": int(rounds), "iterate_and_store": float(a[0]), "poll": float(a[1]),
            "apply_and_barrier": float(a[2]), "poll_passes_per_round_wave0": float(a[3]), "round_us_from_solve": solve_us / rounds,
            "iterate_frac_of_round": float(a[0] / max(a[:3].sum(), 1e-9)),
            "note": "one launch per solve: no kernel boundary, no reload; a round ends with the tile's results stored "
                    "into uncached hand-off copies tagged with the round, then a poll of the neighbours' entries"}


def config_line(name, device, budget_s=2.0):
    """One BASELINE configuration beside the headline one (verdict r03 item 3): resident graph, library defaults,
    device time of the median of up to 9 solves within `budget_s`; contract roofline fraction, the share of a round
    that iterates, the measured HBM share when a PMC summary of these sources is committed."""
    from flame_ros_amd import graphgen
    from flame_ros_amd.regularizer import GraphRegularizer, default_params
    g, iters = graphgen.named(name)
    pr = default_params()
    out = {"V": g.V, "E": g.E, "iters": iters}
    try:
        with GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=device) as r:
            r.step(pr, iters); r.step(pr, iters)
            ms, t0 = [], time.perf_counter()
            while len(ms) < 9 and (len(ms) < 3 or time.perf_counter() - t0 < budget_s):
                r.step(pr, iters)
                ms.append(r.last_solve_ms()[0])
            launches = r.last_solve_ms()[1]
            ms.sort()
            med = ms[len(ms) // 2]
            alg = (84 * g.E + 60 * g.V) * iters
            out.update({"iterations_per_s": iters / (med * 1e-3), "us_per_iteration": med * 1e3 / iters,
                        "launches_per_solve": launches, "resident_tiles": bool(r.info("persist_used")),
                        "num_tiles": r.info("num_tiles"), "tile_depth": r.info("tile_depth"),
                        "contract_frac": alg / (med * 1e-3) / 1e9 / HBM_PEAK_GBPS})
            resident = bool(r.info("persist_used"))
            kern = "k_tile_persist<" if resident else "k_tile<"
            tr = profiled_counters(name, kern)
            counters = None
            if tr and not tr.get("stale"):
                if tr.get("bytes_per_launch"):
                    per_solve = tr["bytes_per_launch"] * launches
                    out["measured_hbm_frac"] = per_solve / (med * 1e-3) / 1e9 / HBM_PEAK_GBPS
                    out["traffic_source"] = tr["source"]
                counters = tr.get("lds")
            if resident:
                sp = resident_round_split(g, iters, device, {}, med * 1e3)
                if sp:
                    out["iterate_frac"] = sp["iterate_frac_of_round"]
                    out["round_split_us"] = [sp["iterate_and_store"], sp["poll"], sp["apply_and_barrier"]]
            else:
                sp = tile_phase_split(g, iters, device, {}, med * 1e3 / max(launches, 1))
                if sp:
                    out["iterate_frac"] = sp["iterate_frac_of_launch"]
            blk = lds_block(r, med * 1e3 / iters, iters / max(launches, 1), counters, out.get("iterate_frac"),
                            out.get("measured_hbm_frac"), out["contract_frac"])
            if blk:  # the same keys as the headline line's roofline block
                out["roofline"] = {k: blk[k] for k in ("bound", "frac", "floor_cycles_per_iteration",
                                                        "measured_cycles_per_iteration", "work_redundancy", "lds_busy",
                                                        "bank_conflict_share", "handoff_share", "measured_hbm_frac",
                                                        "contract_frac") if k in blk}
    except Exception as e:  # noqa: BLE001 -- a side measurement
        out["error"] = str(e)[:200]
    return out


def facade_frames(workloads=("tum", "euroc", "50k")):
    """Median flame::Flame::update latency (C++, tools/facade_bench.cc) of a frame stream with the
    reference's default parameters (cfg/flame_offline_tum.yaml:19-99, debug draws enabled)."""
    sys.path.insert(0, ROOT)
    from tools import facade_bench
    out = {}
    for w in workloads:
        try:
            r = facade_bench.run(w, repeats=10, getters=1)
            out[w] = {"V": r["V"], "iters": r["iters"], "update_ms_p50": r["update_ms"]["p50"],
                      "update_ms_p90": r["update_ms"]["p90"], "sync_graph_ms_p50": r["sync_graph_ms_p50"],
                      "nltgv2_ms_p50": r["nltgv2_ms_p50"], "frames": r["frames"]}
            # update() from features alone: the built-in triangulation runs inside it -- on the GPU (flame_hip_delaunay, the
            # default) and, beside it, on the host pool (Params::triangulate_on_gpu = false)
            f = facade_bench.run(w, repeats=5 if w != "50k" else 3, getters=1, env={"FLAME_BENCH_FRONTEND": "1"})
            out[w]["from_features_update_ms_p50"] = f["update_ms"]["p50"]
            out[w]["from_features_triangulate_ms_p50"] = f["triangulate_ms_p50"]
            f = facade_bench.run(w, repeats=5 if w != "50k" else 3, getters=1, env={"FLAME_BENCH_FRONTEND": "1", "FLAME_BENCH_TRI_GPU": "0"})
            out[w]["from_features_host_triangulator_update_ms_p50"] = f["update_ms"]["p50"]
            out[w]["from_features_host_triangulator_triangulate_ms_p50"] = f["triangulate_ms_p50"]
        except Exception as e:  # noqa: BLE001 -- the bench line must not die on the side measurement
            out[w] = {"error": str(e)[:200]}
    return out


def frames_axis(device, batch=256, win=16, steps=5):
    """Frames axis: `batch` independent TUM-shaped graphs (640x480, one feature per win x win cell) in
    ONE handle, one LDS-resident tile (one CU) per frame, 200 PD iterations each per step."""
    from flame_ros_amd import graphgen
    from flame_ros_amd.regularizer import GraphRegularizer, default_params
    frames = [graphgen.dataset_shaped(640, 480, win, seed=b) for b in range(batch)]
    with GraphRegularizer.from_batch(frames, device=device) as rb:
        pr = default_params()
        for _ in range(3):
            rb.step(pr, 200, sync=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            rb.step(pr, 200, sync=False)
        rb.sync()
        dt = time.perf_counter() - t0
    return {"batch": batch, "win": win, "vertices_per_frame": frames[0].V, "iters": 200,
            "frames_per_s": batch * steps / dt, "frame_iterations_per_s": batch * steps * 200 / dt,
            "r01_frame_iterations_per_s": 1.01e8,
            "note": "one workgroup (one CU) per frame, all 200 iterations in one launch"}


def small_graphs(device):
    """Resident small graphs (TUM-shaped 1.2 k, 5 k, EuRoC-shaped 10 k): microseconds per PD iteration by launches of
    `depth` iterations (persist = 0) and by one launch of resident tiles (the default), device time of the best of 8
    solves each.  r02 / r03 verdict targets: <= 0.9 at 1.2 k, <= 1.0 (r03: <= 1.2) at 5 k and 10 k."""
    from flame_ros_amd import graphgen
    from flame_ros_amd.regularizer import GraphRegularizer, default_params
    pr = default_params()
    out = {}
    for name in ("tum", "5k", "euroc"):
        g, iters = graphgen.named(name)
        o = {"vertices": g.V, "iters": iters}
        for key, kw in (("launches", dict(persist=0)), ("resident", {})):
            try:
                with GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=device, **kw) as r:
                    best = 1e9
                    for _ in range(8):
                        r.step(pr, iters)
                        best = min(best, r.last_solve_ms()[0])
                    o[key] = {"us_per_iteration": best * 1e3 / iters, "tiles": r.info("num_tiles"), "depth": r.info("tile_depth"),
                              "persist_used": r.info("persist_used")}
            except Exception as e:  # noqa: BLE001 -- a side measurement
                o[key] = {"error": str(e)[:200]}
        out[name] = o
    return out


def library_partition(rank, world, device, uid, barrier, max_over_ranks, workload=None, parts_per_rank=1, halo_depth=16,
                      steps=5, pipeline=None, transport=0):
    """BASELINE configs 4 / 5 through the LIBRARY's partition mode (include/flame_hip.h flame_hip_comm_* / flame_hip_part_*,
    csrc/part.cpp: RCB cut, resident tiles per part, ncclSend / ncclRecv halo records on the solve stream) -- not the torch
    harness: ONE graph strong-scaled over world x parts_per_rank subdomains.  50 k vertices below 8 subdomains, 200 k from
    8 on.  Timed: `steps` solves of the config's iterations between barriers, max over ranks.  Beside it the device time of
    an exchange (HIP events inside the library), the bytes a rank moves per exchange, what RCCL says the communicator's
    size is, and the bits of the gathered solution against ONE handle solving the whole graph on this rank's GPU."""
    import numpy as np
    from flame_ros_amd import graphgen, partition
    from flame_ros_amd.regularizer import GraphRegularizer, default_params
    nparts = world * parts_per_rank
    name = workload or ("200k" if nparts >= 8 else "50k")
    g, iters = graphgen.named(name, seed=0)
    p = default_params()
    out = {"workload": name, "V": g.V, "E": g.E, "iters_per_step": iters, "ranks": world, "parts_per_rank": parts_per_rank,
           "halo_depth": halo_depth, "steps": steps, "api": "flame_hip_comm_* / flame_hip_part_* (csrc/part.cpp)"}
    with partition.Communicator(device, rank, world, uid) as comm:
        out["rccl_ranks"] = comm.info("rccl_ranks")
        out["ranks_share_a_gpu"] = bool(comm.info("shared_gpu"))
        with partition.Partition(comm, g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, parts_per_rank=parts_per_rank,
                                 halo_depth=halo_depth) as ps:
            if pipeline is not None:
                ps.set_option("pipeline", int(pipeline))
            if transport:  # r06: the peer transport (records written straight into the receivers' inboxes, 2 launches per exchange)
                ps.set_option("transport", int(transport))
            out["transport"] = "peer" if transport else "rccl"
            # parity first (fresh state): the partitioned solve against one handle on the whole graph
            ps.step(p, iters)
            x, w1, w2, q = ps.gather_solution()
            if rank == 0:
                with GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, device=device) as one:
                    one.step(p, iters)
                    x1, _, _, q1 = one.download()
                out["bit_exact_vs_one_gpu"] = bool(np.array_equal(x.view(np.uint32), x1.view(np.uint32)) and
                                                   np.array_equal(q.view(np.uint32), q1.view(np.uint32)))
            ps.step(p, iters)  # (the second solve of a plan applies the lane order)
            ps.sync()
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                ps.step(p, iters)
            ps.sync()
            barrier()
            dt = max_over_ranks(time.perf_counter() - t0)
            ex0 = ps.info("exchanges")
            ps.set_option("time_exchanges", 1)
            ps.step(p, iters)
            ps.sync()
            out.update({"iterations_per_s": steps * iters / dt, "us_per_iteration": dt / (steps * iters) * 1e6,
                        "ms_per_step": dt / steps * 1e3,
                        "exchanges_per_step": ps.info("exchanges") - ex0, "exchange_us": ps.info("exchange_ns") / 1e3,
                        "p2p_ops_per_exchange_rank0": ps.info("p2p_ops"),
                        "send_bytes_per_exchange_rank0": sum(ps.info("send_bytes", i) for i in range(parts_per_rank)),
                        "recv_bytes_per_exchange_rank0": sum(ps.info("recv_bytes", i) for i in range(parts_per_rank)),
                        "own_vertices_rank0": sum(ps.info("n_own", i) for i in range(parts_per_rank)),
                        "local_vertices_rank0": sum(ps.info("n_ext", i) for i in range(parts_per_rank)),
                        "resident_tiles": bool(ps.info("persist_launches", 0) > 0),
                        "exchanges_pipelined": ps.info("exchanges_pipelined"),
                        "solves_repeated_after_a_give_up": ps.info("recovered")})
            out["exchange_share"] = out["exchange_us"] * out["exchanges_per_step"] / max(out["ms_per_step"] * 1e3, 1e-9)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0,
                    help="timed steps (0 = as many as make the timed window >= 0.25 s, at least 20)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="50k", choices=["5k", "50k", "200k", "tum", "euroc"])
    ap.add_argument("--iters", type=int, default=0, help="PD iterations per step (0 = config)")
    ap.add_argument("--path", type=int, default=0)
    ap.add_argument("--tile-own", type=int, default=0)
    ap.add_argument("--tile-depth", type=int, default=0)
    ap.add_argument("--tile-threads", type=int, default=0)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-balance", action="store_true")
    ap.add_argument("--order-mode", type=int, default=-1)
    ap.add_argument("--host-plan", action="store_true", help="build the plan with the host builder")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT",
                    help="extra library option (flame_hip_graph_set_option), repeatable")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-facade", action="store_true", help="skip the facade frame-latency and frames-axis side measurements")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not measure roofline.traffic by two rocprofv3 PMC child runs (quote the committed profile)")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--mode", default="replicas", choices=["replicas", "partition"],
                    help="N>1: independent frames per GPU (default) or ONE graph cut into N "
                         "subdomains with RCCL halo exchange")
    ap.add_argument("--halo-depth", type=int, default=16)
    ap.add_argument("--partition-timeout", type=int, default=420, help="seconds the library-partition block may take (N>1)")
    ap.add_argument("--no-partition", action="store_true", help="N>1: skip the library-partition block beside the replicas line")
    ap.add_argument("--batch", type=int, default=0,
                    help="frames axis: B independent feature-grid graphs (640x480, one feature per "
                         "--batch-win cell) per step in ONE handle, one LDS tile per frame")
    ap.add_argument("--batch-win", type=int, default=16)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    # FLAME_BENCH_BACKEND=gloo: several ranks on ONE GPU (development check of the N>1 code paths;
    # timing tensors then live on the host); the product backend is nccl (= RCCL over xGMI)
    backend = os.environ.get("FLAME_BENCH_BACKEND", "nccl")
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback"
    ndev = torch.cuda.device_count()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` on its own: start the N ranks here (one process per GPU, the same
        # command line the driver uses) instead of quietly running one
        if ndev < args.gpus and backend != "gloo":
            sys.exit("bench.py --gpus %d: only %d GPU(s) visible (FLAME_BENCH_BACKEND=gloo runs the ranks on "
                     "one GPU as a development check)" % (args.gpus, ndev))
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1:  # (an explicit --gpus must agree with the launcher)
            sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
        args.gpus = world
    shared_gpu = False
    if local_rank >= ndev:  # fewer devices than ranks: only as the gloo development check, all ranks on the devices there are
        if backend != "gloo":
            sys.exit("bench.py: rank %d has no GPU of its own (%d visible); RCCL needs one device per rank" % (rank, ndev))
        local_rank %= ndev
        shared_gpu = True
    shared_gpu = shared_gpu or (backend == "gloo" and world > ndev)
    torch.cuda.set_device(local_rank)
    # FLAME_BENCH_FORCE_PARTITION: run the N > 1 library-partition block at world 1 too (tests: the glue around it -- process
    # group, unique id over the store, watchdog -- on the one GPU there is)
    force_part = bool(os.environ.get("FLAME_BENCH_FORCE_PARTITION")) and world == 1
    if force_part:
        import socket
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(so.getsockname()[1])
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_part:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from flame_ros_amd import graphgen
    from flame_ros_amd.regularizer import GraphRegularizer, default_params

    partition = args.mode == "partition" and world > 1
    if args.batch:
        frames = [graphgen.dataset_shaped(640, 480, args.batch_win, seed=1000 * rank + b)
                  for b in range(args.batch)]

        class _G:  # concatenated view, for the byte/size accounting below
            V = sum(f.V for f in frames)
            E = sum(f.E for f in frames)
        g, cfg_iters = _G, 200
    else:
        g, cfg_iters = graphgen.named(args.workload, seed=0 if partition else rank)
    iters = args.iters or cfg_iters
    opts = {}
    if args.path: opts["path"] = args.path
    if args.tile_own: opts["tile_own"] = args.tile_own
    if args.tile_depth: opts["tile_depth"] = args.tile_depth
    if args.tile_threads: opts["tile_threads"] = args.tile_threads
    if args.no_graph: opts["use_graph"] = 0
    if args.no_balance: opts["balance"] = 0
    if args.host_plan: opts["plan_device"] = 0
    if args.order_mode >= 0: opts["order_mode"] = args.order_mode
    if shared_gpu:  # resident tiles assume the whole chip; ranks sharing one GPU would starve each other
        opts["persist"] = 0
    for kv in args.opt:
        k, v = kv.split("=")
        opts[k] = int(v)
    p = default_params()
    if partition:
        from flame_ros_amd import dist as fdist
        ps = fdist.PartitionedSolver(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt,
                                     fdist.make_hip_solver(local_rank, **opts), depth=args.halo_depth)

        class _R:  # same surface as GraphRegularizer for the timing loop below
            def step(self, p, n, sync=False):
                ps.step(p, n)

            def sync(self):
                torch.cuda.synchronize()

            def last_solve_ms(self):
                return float("nan"), 0

            def info(self, k):
                return ps.solver.reg.info(k)
        r = _R()
    elif args.batch:
        r = GraphRegularizer.from_batch(frames, device=local_rank, **opts)
    else:
        r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris,
                             device=local_rank, **opts)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        r.sync()

    # setup, not a warmup step: the library replays a hipGraph from the second solve after an
    # upload on (a frame stream that re-uploads every frame never pays capture + instantiate), so
    # two priming solves put capture/instantiate outside the timed region whatever --warmup is
    for _ in range(2):
        r.step(p, iters, sync=True)
    if args.steps <= 0:  # default: a timed window of >= 0.25 s (r02's 20 steps were a 20 ms window)
        r.sync()
        te = time.perf_counter()
        for _ in range(5):
            r.step(p, iters, sync=False)
        r.sync()
        step_s = (time.perf_counter() - te) / 5
        args.steps = int(max(20, min(5000, -(-0.25 // step_s))))
        if world > 1:  # every rank must time the same number of steps
            t = torch.tensor([args.steps], device="cuda" if backend == "nccl" else "cpu", dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            args.steps = int(t.item())
    for _ in range(args.warmup):
        r.step(p, iters, sync=False)
    barrier()
    launches = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r.step(p, iters, sync=False)
    r.sync()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the contract's figure is the ONE window above; it is short (tens of ms), so the same K-step
    # window is repeated a few times afterwards and the spread reported beside it (rank 0 only)
    repeat_ips = []
    if world == 1:
        for _ in range(5):
            r.sync()
            tr0 = time.perf_counter()
            for _ in range(args.steps):
                r.step(p, iters, sync=False)
            r.sync()
            repeat_ips.append(args.steps * iters * max(args.batch, 1) / (time.perf_counter() - tr0))
        repeat_ips.sort()

    # per-launch device time with HIP events on the solve stream (outside the timed region so the
    # event pairs do not perturb it): same launches, one solve at a time
    ev_ms = []
    for _ in range(min(args.steps, 10)):
        r.step(p, iters, sync=True)
        ms, launches = r.last_solve_ms()
        ev_ms.append(ms)
    ev_ms.sort()
    solve_ms = ev_ms[len(ev_ms) // 2]
    if solve_ms != solve_ms:  # partition mode: no single-handle event pair; use the wall clock
        solve_ms, launches = elapsed / args.steps * 1e3, max(1, -(-iters // max(1, r.info("tile_depth") or iters)))

    # PCIe-inclusive frame rate (never `value`): host graph in -> plan build + H2D -> iterations
    # -> D2H of x and the triangle stage, a fresh handle per frame as a real frame stream would do
    frame_ms = None
    if not partition and not args.batch:
        ts = []
        import ctypes as C
        rf = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris,
                              device=local_rank, **opts)
        pp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        for _ in range(14):  # one persistent handle, re-uploaded per frame (streaming use)
            t0f = time.perf_counter()
            rc = rf._lib.flame_hip_graph_upload(rf._h, pp(g.pos), pp(g.edges), pp(g.alpha), pp(g.beta),
                                                pp(g.z), pp(g.wgt), None, pp(g.tris))
            assert rc == 0, rc
            rf.step(p, iters, sync=False)
            rf.download(with_q=False)
            ts.append((time.perf_counter() - t0f) * 1e3)
        rf.close()
        ts = sorted(ts[2:])  # (frame 1 builds the partition, frame 2 is the first to reuse it)
        frame_ms = ts[len(ts) // 2]

    part_info = None
    if partition:  # per-exchange cost, measured apart from the timed region: pack + P2P + unpack
        torch.cuda.synchronize(); dist.barrier()
        t0x = time.perf_counter()
        for _ in range(20):
            ps.exchange()
        torch.cuda.synchronize(); dist.barrier()
        ex_us = (time.perf_counter() - t0x) / 20 * 1e6
        sb, rb = 4 * (6 * ps.n_send[0] + 3 * ps.n_send[1]), 4 * (6 * ps.n_recv[0] + 3 * ps.n_recv[1])
        part_info = {"halo_depth": args.halo_depth, "exchanges_per_step": max(0, -(-iters // args.halo_depth) - 1),
                     "send_bytes_rank0": sb, "recv_bytes_rank0": rb, "peers_rank0": len(ps.peers),
                     "exchange_us": ex_us, "own_vertices_rank0": int(ps.sub.n_own),
                     "halo_vertices_rank0": int(len(ps.sub.vid) - ps.sub.n_own)}
    if rank == 0:
        nfr = max(args.batch, 1)
        total_iters = (1 if partition else world) * args.steps * iters * nfr
        alg_bytes_iter = 84 * g.E + 60 * g.V
        path = r.info("path")
        resident = bool(r.info("persist_used"))
        iters_per_launch = iters / max(launches, 1)
        launch_us = solve_ms * 1e3 / max(launches, 1)
        achieved = alg_bytes_iter * iters_per_launch / (launch_us * 1e-6) / 1e9
        out = {
            "metric": "primal-dual iterations/sec on a %d-vertex / %d-edge Delaunay graph" % (g.V, g.E),
            "value": total_iters / elapsed, "unit": "PD iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if partition else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("synthetic %s-vertex Delaunay graph" % args.workload if args.workload[0].isdigit()
                                    else "%s-shaped feature-grid graph (%d vertices)" % (args.workload, g.V)) +
                                   ", %d PD iterations per frame, one frame per GPU" % iters,
                       "V": g.V, "E": g.E, "iters_per_step": iters, "parallelism": ("partition%d_halo%d" % (world, args.halo_depth)) if partition else "replicas%d" % world,
                       "path": {1: "global", 2: "tile"}[path], "num_tiles": r.info("num_tiles"),
                       "tile_depth": r.info("tile_depth"), "tile_threads": r.info("tile_threads"),
                       "hipgraph": not args.no_graph,
                       # resident launches that gave up and were repeated by launches (whole queues included, r05) while this
                       # handle was measured: normally 0; > 0 means some window paid for a repeat -- the value stays valid
                       "resident_solves_repeated": r.info("persist_recovered") if not partition and not args.batch else None},
            "repeats": ({"windows": len(repeat_ips), "median": repeat_ips[len(repeat_ips) // 2], "min": repeat_ips[0],
                         "max": repeat_ips[-1], "unit": "PD iterations/s",
                         "note": "the same K-step window repeated after the timed one"} if repeat_ips else None),
            "frames_per_s": (1 if partition else world) * args.steps * nfr / elapsed,
            "us_per_iteration": elapsed / (args.steps * iters) * 1e6,
            "roofline": {"bound": "hbm", "contract_bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "kernel": ("k_tile_persist" if resident else "k_tile") if path == 2 else "k_dual+k_primal",
                         "launch_us": launch_us, "iters_per_launch": iters_per_launch,
                         "alg_bytes_per_iter": alg_bytes_iter,
                         "note": "global path: achieved / frac = (84E+60V) x iterations per launch / mean launch duration "
                                 "(HIP events on the solve stream, incl. launch gaps) against the HBM peak (SURVEY 8d)."},
        }
        if part_info:
            out["partition_torch_harness"] = part_info
        if frame_ms:
            out["host_inclusive"] = {"ms_per_frame": frame_ms, "frames_per_s": 1e3 / frame_ms,
                                     "iterations_per_s": iters * 1e3 / frame_ms,
                                     "plan_on_device": bool(r.info("plan_on_device")),
                                     "note": "host arrays in -> H2D + plan build (on the GPU when plan_on_device) + solve + "
                                             "D2H, one handle re-uploaded per frame; informational, never `value`"}
        rl = out["roofline"]
        if path == 2 and not args.batch:
            # what a temporally blocked launch has to move at least: every tile reads its local
            # state + constants once and writes what it owns (per-iteration algorithmic bytes do
            # not apply to a kernel that keeps d iterations on chip)
            nv, ne = r.info("tile_ext_vertices"), r.info("tile_loc_edges")
            floor = nv * (32 + 8) + ne * (16 + 28) + g.V * 32 + g.E * 16
            if resident:  # one launch per solve: that once, plus per hand-off what the tiles own (out) and their halos (in)
                rounds = -(-iters // max(r.info("tile_depth"), 1))
                floor += (rounds - 1) * (g.V * 32 + g.E * 16 + (nv - g.V) * 32 + (ne - g.E) * 16)
            rl["blocked_floor_bytes_per_launch"] = floor
            rl["halo_redundancy"] = {"vertices": nv / g.V, "edges": ne / max(g.E, 1)}
        tr = profiled_counters("batch%d" % args.batch if args.batch else args.workload,
                               ("k_tile_persist<" if resident else "k_tile<") if path == 2 else "k_primal")
        kern_sub = ("k_tile_persist<" if resident else "k_tile<") if path == 2 else "k_primal"
        live = None
        if world == 1 and not args.no_live_traffic and not (args.tile_own or args.tile_depth or args.tile_threads or partition or args.opt or args.host_plan):
            # r06 (VERDICT r05 "evidence hygiene"): the line's HBM bytes are THIS run's counters, not a committed profile's
            child = ["--workload", args.workload] + (["--iters", str(args.iters)] if args.iters else []) + \
                    (["--batch", str(args.batch), "--batch-win", str(args.batch_win)] if args.batch else [])
            # (the exact variant: the child's side measurements launch other instantiations of the same template)
            kern_live = ("k_tile_persist<%d, %d, %d," % (r.info("tile_threads"), r.info("tile_ept"), r.info("tile_vpt"))) if (path == 2 and resident) else kern_sub
            live = live_hbm_traffic(kern_live, child)
        if live:  # (an LDS / VALU pass that did not come stays the committed one's -- when that was taken from these sources)
            base = tr if (tr and not tr.get("stale")) else {"lds": None, "valu": None}
            tr = dict(base, bytes_per_launch=live["bytes_per_launch"], source=live["source"], stale=False,
                      lds=live["lds"] or base.get("lds"), valu=live["valu"] or base.get("valu"),
                      committed_source=None if (live["lds"] and live["valu"]) else base.get("source"))
        if tr and not (args.tile_own or args.tile_depth or args.tile_threads or partition):
            rl["traffic_source"] = tr["source"] + (" (stale: other kernel sources, not quoted)" if tr.get("stale") else "")
            if live and tr.get("committed_source"):
                rl["counters_source"] = tr["committed_source"]  # (an LDS / VALU set that was not measured live: the committed passes of the same sources)
            if tr.get("bytes_per_launch"):
                rl["traffic"] = tr["bytes_per_launch"]
                rl["measured_hbm_gbps"] = tr["bytes_per_launch"] / (launch_us * 1e-6) / 1e9
                rl["measured_hbm_frac"] = rl["measured_hbm_gbps"] / HBM_PEAK_GBPS
                if rl.get("blocked_floor_bytes_per_launch"):
                    rl["traffic_over_blocked_floor"] = tr["bytes_per_launch"] / rl["blocked_floor_bytes_per_launch"]
        if path == 2 and not args.batch and not partition:
            if resident:
                rs_ = resident_round_split(g, iters, local_rank, opts, launch_us)
                if rs_:
                    rl["round_split"] = rs_
                    rl["iterate_frac"] = rs_["iterate_frac_of_round"]
            else:
                ps_ = tile_phase_split(g, iters, local_rank, opts, launch_us)
                if ps_:
                    rl["phase_split"] = ps_
                    rl["iterate_frac"] = ps_["iterate_frac_of_launch"]
        if path == 2 and not partition:
            # The tile path keeps the state on chip across the iterations of a launch (resident tiles: of the whole solve),
            # so the contract's HBM figure does not bound it (it read 1.36 at 50 k and 5.6 on the batch line in r04).  The
            # primary roofline is the ON-CHIP one: LDS issue floor of the work the tiles OWN over the measured cycles per
            # iteration (<= 1 by construction); the contract's number stays beside it as contract_frac.
            contract = {"bound": "hbm", "achieved": rl["achieved"], "peak": rl["peak"], "unit": rl["unit"], "frac": rl["frac"],
                        "note": "(84E+60V) x iterations per launch / launch duration / 8 TB/s: SURVEY 8d's figure; not a bound for "
                                "a kernel whose state stays in LDS (may exceed 1)"}
            blk = lds_block(r, launch_us / iters_per_launch, iters_per_launch, (tr or {}).get("lds") if tr and not tr.get("stale") else None,
                            rl.get("iterate_frac"), rl.get("measured_hbm_frac"), contract["frac"],
                            (tr or {}).get("valu") if tr and not tr.get("stale") else None)
            if blk:
                rl.update(blk)
                rl["contract"] = contract
                rl["note"] = ("bound = the larger of two on-chip issue floors of the work the tiles OWN (r06): lds_frac / valu_frac beside frac; "
                              "valu floor = 16 (phase D, per 64 own edges) / 16 + 2 per slot (phase P, per 64 own vertices) wave-instructions "
                              "priced 4 cycles each over 4 SIMDs; contract.frac = SURVEY 8d's HBM figure, NOT a bound: the "
                              "state is LDS-resident.  lds: frac = LDS issue floor of the slowest CU (2 ds_read_b128 + 2 ds_write_b96 per 64 own "
                              "edges, max-degree slot reads + 1 store per 64 own vertices, priced 4 / 10 cycles: MI355X guide LDS "
                              "table) / measured shader cycles per iteration (HIP events on the solve stream).  work_redundancy = "
                              "executed / useful LDS wave-instructions (halo rings, padding lanes); lds_busy = SQ_LDS_IDX_ACTIVE per "
                              "CU / cycles; handoff_share = 1 - iterate_frac (round_split); contract_frac = the SURVEY 8d HBM figure; "
                              "measured_hbm_frac = PMC FETCH x2 + WRITE per launch / time / 8 TB/s.  the counters are measured IN this run when rocprofv3 is there "
                              "(four short child runs under --pmc: FETCH_SIZE, WRITE_SIZE, the LDS set, the VALU / wait set; traffic_source says so); a set "
                              "that did not come is the committed rocprofv3 pass of the same sources (counters_source); profiles/summarize.py --roofline recomputes.")
        if args.batch:
            out["metric"] = "primal-dual iterations/sec over a batch of %d independent %d-vertex graphs" % (
                args.batch, frames[0].V)
            out["config"]["workload"] = "batch of %d feature-grid graphs (640x480, win %d), %d PD iterations each" % (
                args.batch, args.batch_win, iters)
            out["roofline"]["note"] += " Batch mode: value counts frame-iterations; one isolated tile (one CU) per frame."
        if world == 1 and not args.batch and not args.no_facade:
            out["facade_frame_ms"] = facade_frames()
            out["facade_frame_ms"]["note"] = ("median flame::Flame::update of a 40-frame stream through the C++ "
                                              "facade with reference-default params (debug draws enabled), "
                                              "tools/facade_bench.cc; targets 0.6 / 1.0 / 2.2 ms")
            out["frames_axis"] = frames_axis(local_rank)
            out["small_graph_us_per_iteration"] = small_graphs(local_rank)
            out["other_configs"] = {w: config_line(w, local_rank) for w in ("tum", "5k", "euroc", "200k") if w != args.workload}
            out["other_configs"]["note"] = ("BASELINE configs 1 (TUM-shaped, 1.2 k vertices) / 2 / 3 / 5 on one GPU, resident graph, library defaults: device time "
                                            "of the median solve; contract_frac = (84E+60V) x iterations / time / 8 TB/s")
        if not args.no_cpu and world == 1:  # contract: rank 0 at N=1 only
            cb = cpu_baseline(args.workload, args.batch_win if args.batch else 0, iters, args.cpu_budget)
            out["cpu_baseline"] = cb
            if not args.batch:
                out["parity_vs_oracle"] = parity_check(g, iters, local_rank, opts)
            out["speedup_vs_cpu_1thread"] = out["value"] / (1 if partition else world) / cb["value"]
            out["speedup_vs_cpu_best"] = out["value"] / (1 if partition else world) / max(cb["best_value"], cb["value"])
    else:
        out = None
    lib_part = None
    if (world > 1 or force_part) and backend == "nccl" and not args.batch and not args.no_partition and not partition:  # (gloo development runs: ranks share a GPU)
        # the LIBRARY's partition mode beside the replicas line: the unique id travels over the torch
        # store, everything else is flame_hip_comm_* / flame_hip_part_* -- RCCL by the library itself, on its own stream
        from flame_ros_amd import partition as fpart
        # A collective that never completes (a rank that died, an RCCL mismatch) must not cost the replicas line: a watchdog
        # THREAD (the main thread may sit in a C call for good) prints the line without the block and ends the process.
        import threading

        def _bail():
            if rank == 0:
                out["partition"] = {"error": "the library-partition block did not finish within %d s" % args.partition_timeout}
                print(json.dumps(out), flush=True)
            os._exit(0)
        dog = threading.Timer(args.partition_timeout, _bail)
        dog.daemon = True
        dog.start()
        box = [fpart.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)

        def _max(v):
            t = torch.tensor([v], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        def _bar():
            torch.cuda.synchronize()
            dist.barrier()
        try:
            lib_part = library_partition(rank, world, local_rank, box[0], _bar, _max, halo_depth=args.halo_depth)
        except Exception as e:  # noqa: BLE001 -- the replicas line (the contract's `value`) must survive a failure of the side block
            lib_part = {"error": "%s: %s" % (type(e).__name__, str(e)[:300]), "rccl_ranks": world}
        if lib_part["rccl_ranks"] != world:
            sys.exit("bench.py: RCCL reports %d ranks in the library's communicator, launched with %d" % (lib_part["rccl_ranks"], world))
        try:  # a deeper halo: half the exchanges for more redundant work (the parts stay resident up to ~49 k local vertices)
            box3 = [fpart.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box3, src=0)
            dh = library_partition(rank, world, local_rank, box3[0], _bar, _max, workload=lib_part.get("workload"),
                                   halo_depth=2 * args.halo_depth, steps=3)
            lib_part["halo_depth_x2"] = {k: dh[k] for k in ("halo_depth", "iterations_per_s", "us_per_iteration", "exchanges_per_step",
                                                            "exchange_us", "bit_exact_vs_one_gpu", "resident_tiles", "local_vertices_rank0")
                                         if k in dh}
        except Exception as e:  # noqa: BLE001
            lib_part["halo_depth_x2"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        try:  # r06: the same cut through the PEER transport (two launches per exchange, no ncclGroup): RCCL's figures stay beside it
            box4 = [fpart.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box4, src=0)
            pt = library_partition(rank, world, local_rank, box4[0], _bar, _max, workload=lib_part.get("workload"),
                                   halo_depth=args.halo_depth, steps=3, transport=1)
            lib_part["peer_transport"] = {k: pt[k] for k in ("transport", "halo_depth", "iterations_per_s", "us_per_iteration", "exchanges_per_step",
                                                             "exchange_us", "exchange_share", "bit_exact_vs_one_gpu", "resident_tiles")
                                          if k in pt}
        except Exception as e:  # noqa: BLE001
            lib_part["peer_transport"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        try:  # the same graph over-decomposed, two parts per rank: the records of part 0 travel while part 1 iterates
            box2 = [fpart.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box2, src=0)
            od = library_partition(rank, world, local_rank, box2[0], _bar, _max, workload=lib_part.get("workload"),
                                   parts_per_rank=2, halo_depth=args.halo_depth, steps=3, pipeline=1)
            lib_part["two_parts_per_rank_pipelined"] = {k: od[k] for k in ("iterations_per_s", "us_per_iteration", "exchanges_per_step",
                                                                         "exchanges_pipelined", "bit_exact_vs_one_gpu", "resident_tiles")
                                                        if k in od}
        except Exception as e:  # noqa: BLE001
            lib_part["two_parts_per_rank_pipelined"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

        dog.cancel()
    if rank == 0:
        if lib_part:
            out["partition"] = lib_part
        print(json.dumps(out))
    if world > 1 or force_part:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
