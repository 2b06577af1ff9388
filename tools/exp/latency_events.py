"""r06: what makes a resident solve LATE on a GPU that is otherwise quiet?  One handle, solve after solve
(synchronised each), device time of every solve (HIP events around the launch) and the longest poll wait inside it, while a
second host thread of the SAME process does one kind of thing: nothing / hipMalloc + hipFree / page-locked allocations /
64 MB copies on another stream / tiny kernels on another stream.  Then the same on a fat plan that fills a CU's LDS to the
last KiB (option lds_bytes raised past the library's margin).
  python tools/exp/latency_events.py [solves]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from flame_ros_amd import graphgen  # noqa: E402
from flame_ros_amd.regularizer import GraphRegularizer, default_params  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
p = default_params()


def disturb(kind, stop):
    st = torch.cuda.Stream()
    if kind == "malloc_free":
        while not stop.is_set():
            t = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
            del t
            torch.cuda.empty_cache()  # (hipFree: the caching allocator would keep the block otherwise)
    elif kind == "pinned_alloc":
        while not stop.is_set():
            t = torch.empty(8 << 20, dtype=torch.uint8, pin_memory=True)
            del t
    elif kind == "copies":
        h = torch.empty(64 << 20, dtype=torch.uint8, pin_memory=True)
        d = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
        with torch.cuda.stream(st):
            while not stop.is_set():
                d.copy_(h, non_blocking=True)
                h.copy_(d, non_blocking=True)
                st.synchronize()
    elif kind == "tiny_kernels":
        a = torch.zeros(4096, device="cuda")
        with torch.cuda.stream(st):
            while not stop.is_set():
                for _ in range(64):
                    a.add_(1.0)
                st.synchronize()


def run(r, iters, n, kind):
    stop = threading.Event()
    th = None
    if kind != "quiet":
        th = threading.Thread(target=disturb, args=(kind, stop))
        th.start()
        time.sleep(0.05)
    dev, wall, waits = [], [], []
    g0, rec0 = r.info("persist_gave_up"), r.info("persist_recovered")
    try:
        for _ in range(n):
            t0 = time.perf_counter()
            r.step(p, iters)
            wall.append((time.perf_counter() - t0) * 1e3)
            dev.append(r.last_solve_ms()[0])
            waits.append(r.info("persist_wait_us_max"))
    finally:
        stop.set()
        if th:
            th.join()
    d, w = np.sort(np.asarray(dev)), np.sort(np.asarray(wall))
    slow = [(i, round(dev[i], 3), waits[i]) for i in range(n) if dev[i] > 5.0 * d[n // 2]][:6]
    print("  %-13s device ms p50 %.3f p99 %.3f max %.3f | host ms p50 %.3f p99 %.3f max %.3f | poll wait us p50 %d max %d | gave up %d repeated %d resident %d%s" % (
        kind, d[n // 2], d[int(0.99 * (n - 1))], d[-1], w[n // 2], w[int(0.99 * (n - 1))], w[-1], int(np.median(waits)), max(waits),
        r.info("persist_gave_up") - g0, r.info("persist_recovered") - rec0, r.info("persist_used"),
        (" | slow solves (index, ms, wait us): %s" % slow) if slow else ""), flush=True)


for name, V, iters, opts in (("50k regular tiles", 50000, 100, {}),
                             ("190k fat tiles, library margin", 190000, 100, {}),
                             ("190k fat tiles, LDS filled to the limit", 190000, 100, dict(lds_bytes=160 * 1024 + 12 * 1024))):
    g = graphgen.synthetic(V, 1280 if V > 60000 else 640, 1024 if V > 60000 else 480, seed=4)
    with GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, device=0, **opts) as r:
        r.step(p, iters); r.step(p, iters)
        print("== %s: %d tiles depth %d, LDS %d B per tile, slot12 %d" % (name, r.info("num_tiles"), r.info("tile_depth"), r.info("tile_lds_bytes"), r.info("tile_slot12")), flush=True)
        for kind in ("quiet", "malloc_free", "pinned_alloc", "copies", "tiny_kernels", "quiet"):
            run(r, iters, N if V < 60000 else N // 3, kind)
