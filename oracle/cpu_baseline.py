"""CPU baseline leg of bench.py (TEST INFRASTRUCTURE, see oracle/__init__.py): the oracle timed on
the host cores of the box it runs on, in its OWN process so that

  * the OpenMP runtime starts with pinned threads (OMP_PROC_BIND=close, OMP_PLACES=cores must be
    in the environment before libgomp initialises; inside bench.py torch has already loaded one),
  * the timed library is compiled -march=native for THIS host (oracle/_native/, git-ignored); the
    library the parity tests compare with stays the portable x86-64-v3 build: both are
    -ffp-contract=off and bit-identical (checked here before timing).

Prints one JSON object: 1-thread rate (the contract's figure: upstream's step() is a sequential
graph walk), the best of {1,4,8,16,32,...} threads of the OpenMP variant, CPU model, core count.

  OMP_PROC_BIND=close OMP_PLACES=cores python -m oracle.cpu_baseline --workload 50k --budget 12
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build_native():
    out_dir = os.path.join(HERE, "_native")
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, "libnltgv2_oracle_native.so")
    src = os.path.join(HERE, "nltgv2_oracle.c")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-march=native", "-ffp-contract=off", "-fopenmp", "-fPIC",
                               "-std=c11", "-shared", "-o", lib, src, "-lm"])
    return lib


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cgroup_cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max, v1 cfs quota / period), or None when unlimited: the
    OpenMP legs cannot scale past it whatever `host_cores` says (r05: best at 8 threads under a 16-CPU quota on a 256-core host)."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="50k")
    ap.add_argument("--batch-win", type=int, default=0, help="dataset-shaped 640x480 graph instead")
    ap.add_argument("--iters", type=int, default=0)
    ap.add_argument("--budget", type=float, default=12.0)
    args = ap.parse_args()
    try:  # before any OpenMP runtime loads: with OMP_PROC_BIND the initial thread is then bound to
        ncpu = len(os.sched_getaffinity(0))  # ONE place and the affinity mask shrinks to it
    except AttributeError:
        ncpu = os.cpu_count() or 1
    import numpy as np
    from flame_ros_amd import graphgen
    from oracle import cbind
    if args.batch_win:
        g, iters = graphgen.dataset_shaped(640, 480, args.batch_win, seed=0), 200
    else:
        g, iters = graphgen.named(args.workload, seed=0)
    iters = args.iters or iters
    p = cbind.default_params()
    # portable build (the parity checker) for the bit-identity check, native build for the timing
    ref = cbind.COracle(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt)
    ref.solve(p, 7)
    native = build_native()
    cbind._lib = C.CDLL(native)
    cbind._lib.nltgv2_solve.restype = C.c_int
    o = cbind.COracle(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt)
    o.solve(p, 7)
    assert np.array_equal(o.x.view(np.uint32), ref.x.view(np.uint32)), "native build differs from the portable oracle"
    chunk = max(10, iters // 10)

    def timed(fn, budget):
        fn(5)
        done, t0 = 0, time.perf_counter()
        while True:
            fn(chunk)
            done += chunk
            dt = time.perf_counter() - t0
            if dt >= budget or done >= 20 * iters:
                return done / dt, done

    one, done = timed(lambda n: o.solve(p, n), args.budget)
    threads = {}
    legs = [t for t in (1, 4, 8, 16, 32, 64) if t <= ncpu]
    for nt in legs:
        o2 = cbind.COracle(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt)
        rate, _ = timed(lambda n: o2.solve_threads(p, n, nt), max(1.0, args.budget / 5))
        threads[str(nt)] = rate
    o2 = cbind.COracle(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt)
    o2.solve_threads(p, 7, legs[-1])
    assert np.array_equal(o2.x.view(np.uint32), ref.x.view(np.uint32)), "OpenMP variant differs from the oracle"
    best_t = max(threads, key=lambda k: threads[k])
    print(json.dumps({
        "value": one, "unit": "PD iterations/s", "cores": 1, "kind": "port",
        "sample": "%d PD iterations of the same %d-vertex/%d-edge graph, 1 thread, gcc -O3 -march=native "
                  "-ffp-contract=off (oracle/nltgv2_oracle.c)" % (done, g.V, g.E),
        "threads_its_per_s": threads, "best_threads": int(best_t), "best_value": threads[best_t],
        "omp_proc_bind": os.environ.get("OMP_PROC_BIND"), "omp_places": os.environ.get("OMP_PLACES"),
        "host_cores": ncpu, "cgroup_cpu_quota": cgroup_cpu_quota(), "cpu_model": cpu_model(),
        "note": "quote absolutes: the OpenMP legs are bound by the container's CPU quota, not by host_cores"}))


if __name__ == "__main__":
    main()
