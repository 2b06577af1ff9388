"""Latency of the integration path: frame streams through flame::Flame::updateGraph (C++,
tools/facade_bench.cc) with the reference's default parameters at the BASELINE sizes.

  python tools/facade_bench.py [--workloads tum,5k,euroc,50k] [--repeats 25] [--getters 2]

Builds tools/facade_bench in-tree (g++ -std=c++11, links libflame_hip.so), writes 4 distinct frames
per workload (graphgen seeds 0..3; every frame is a new graph to the library) into a temp dir, runs
the stream and prints one JSON object {workload: {...}}.  bench.py imports `run()` for its
`facade_frame_ms` block."""
import argparse
import json
import os
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EXE = os.path.join(ROOT, "tools", "facade_bench")
SRC = os.path.join(ROOT, "tools", "facade_bench.cc")


def build(force=False):
    from flame_ros_amd import lib
    lib.load()
    deps = [SRC] + [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(ROOT, "include")) for f in fs]
    if not force and os.path.exists(EXE) and all(os.path.getmtime(d) <= os.path.getmtime(EXE) for d in deps):
        return EXE
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"), SRC,
                           "-o", EXE, "-L" + os.path.join(ROOT, "flame_ros_amd"), "-lflame_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "flame_ros_amd"), "-pthread"])
    return EXE


def write_frame(path, g, iters):
    import numpy as np
    with open(path, "wb") as f:
        f.write(struct.pack("<5i", g.V, g.T, iters, 0, 0))
        f.write(g.pos.astype(np.float32).tobytes())
        f.write(g.z.astype(np.float32).tobytes())
        f.write(g.tris.astype(np.int32).tobytes())


def run(workload, repeats=25, getters=1, nframes=4, env=None):
    from flame_ros_amd import graphgen
    exe = build()
    with tempfile.TemporaryDirectory() as td:
        files, g0, iters = [], None, 0
        for k in range(nframes):
            g, iters = graphgen.named(workload, seed=k)
            g0 = g0 or g
            p = os.path.join(td, "f%d.bin" % k)
            write_frame(p, g, iters)
            files.append(p)
        out = subprocess.run([exe, str(g0.width), str(g0.height), str(iters), str(repeats), str(getters)] + files,
                             capture_output=True, text=True, env=dict(os.environ, **(env or {})), timeout=900)
    if out.returncode != 0:
        raise RuntimeError("facade_bench %s failed (%d): %s %s" % (workload, out.returncode, out.stdout[-500:], out.stderr[-500:]))
    if os.environ.get("FLAME_BENCH_SERIES"):  # dev: the per-frame series goes to stderr
        sys.stderr.write(out.stderr[-8000:])
    return json.loads(out.stdout.strip().splitlines()[-1])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="tum,5k,euroc,50k")
    ap.add_argument("--repeats", type=int, default=25)
    ap.add_argument("--getters", type=int, default=1)
    ap.add_argument("--no-debug", action="store_true", help="A/B: debug draws off")
    a = ap.parse_args()
    res = {}
    for w in a.workloads.split(","):
        res[w] = run(w, a.repeats, a.getters, env={"FLAME_BENCH_NO_DEBUG": "1"} if a.no_debug else None)
    print(json.dumps(res))
