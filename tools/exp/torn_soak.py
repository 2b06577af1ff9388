"""Dev aid: the resident tiles' hand-offs counted for TORN 16-byte entries.  Run with the debug build
(tools/exp/build_variant.sh torn -DFLAME_TORN_CHECK=1; FLAME_HIP_LIB=flame_ros_amd/libflame_hip_torn.so): every entry's
tag word carries a hash of its payload, a reader that sees this round's tag beside another round's payload counts it.
50 k vertices on 256 tiles, `n` solves of 500 iterations beside a stream of matmuls (uneven load), every 10th solve
compared bit for bit with a handle solved by launches.  python tools/exp/torn_soak.py [n]"""
import os
import sys
import threading

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flame_ros_amd import graphgen  # noqa: E402
from flame_ros_amd.regularizer import GraphRegularizer, default_params  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
name = sys.argv[2] if len(sys.argv) > 2 else "50k"  # (r05: "200k", "v100000" ... = the FAT variants' hand-offs)
g, it = graphgen.named(name)
p = default_params()
res = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0)
ref = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, persist=0)
stop = threading.Event()


def hog():
    st = torch.cuda.Stream()
    a = torch.randn(2048, 2048, device="cuda")
    b = torch.randn(2048, 2048, device="cuda")
    with torch.cuda.stream(st):
        while not stop.is_set():
            for _ in range(16):
                a @ b
            st.synchronize()


t = threading.Thread(target=hog)
t.start()
used = bad = 0
try:
    for k in range(n):
        res.step(p, it)
        used += res.info("persist_used")
        ref.step(p, it)
        if k % 10 == 9:
            x, w1, w2, q = res.download()
            xr, w1r, w2r, qr = ref.download()
            bad += int((x.view(np.uint32) != xr.view(np.uint32)).sum() + (q.view(np.uint32) != qr.view(np.uint32)).sum())
finally:
    stop.set()
    t.join()
tiles = res.plan_array("tiles", np.int32).reshape(-1, 47)
entries = int((2 * (tiles[:, 2] - tiles[:, 1]) + (tiles[:, 5] - tiles[:, 4])).sum())  # halo entries polled per round (B + A + q, roughly)
rounds = -(-it // res.info("tile_depth")) - 1
print("torn-read debug build: %d | %d resident solves of %d (repeated by launches %d, give-ups %d) | ~%.2e hand-off entries read (%d per round x %d rounds x solves) | TORN entries counted: %d | differing words vs launches: %d | longest poll wait %d us (time-out %d us)" % (
    res.info("torn_check_build"), used, n, res.info("persist_recovered"), res.info("persist_gave_up"), float(entries) * rounds * used, entries, rounds,
    res.info("persist_torn"), bad, res.info("persist_wait_us_max"), res.info("persist_timeout_us")))
