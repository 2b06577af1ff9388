"""r06: the partition mode's PEER transport (include/flame_hip.h "transport" 1; csrc/part.cpp exchange_peer): halo records
written by one kernel of the sending rank straight into the receiving parts' inboxes, a flag word per message, one kernel of
the receiving rank that waits for its flags and unpacks -- beside RCCL's send / receive (the contract's path, unchanged)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

WORLD1 = r'''
import sys
import numpy as np
sys.path.insert(0, %r)
from flame_ros_amd import graphgen, partition
from flame_ros_amd.regularizer import default_params
from oracle import COracle
from oracle.cbind import default_params as oparams
p = default_params()
for use_rccl in (1, 0):  # a communicator over RCCL (world 1: the transport is a choice), then one WITHOUT RCCL at all
    uid = partition.unique_id() if use_rccl else None
    with partition.Communicator(0, 0, 1, uid) as comm:
        for V, k, depth, iters in ((6000, 2, 8, 50), (9000, 3, 4, 23), (50000, 2, 16, 100), (30000, 8, 6, 40)):
            g = graphgen.synthetic(V, seed=11)
            o = COracle(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt)
            o.solve(oparams(), iters)
            with partition.Partition(comm, g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, parts_per_rank=k, halo_depth=depth) as ps:
                ps.set_option("transport", 1)
                assert ps.info("transport") == 1 and ps.info("peer_connected") == 1 and ps.info("inbox_bytes") > 0
                ps.step(p, iters // 2)
                ps.step(p, iters - iters // 2)
                x, w1, w2, q = ps.gather_solution()
                assert ps.info("exchanges") == (iters - 1) // depth and ps.info("peer_epoch") >= ps.info("exchanges") and ps.info("peer_timeouts") == 0
                for name, got, want in (("x", x, o.x), ("w1", w1, o.w1), ("w2", w2, o.w2), ("q", q, o.q)):
                    assert np.array_equal(got.view(np.uint32), np.asarray(want, np.float32).view(np.uint32)), (use_rccl, V, k, name)
                sm, da = ps.costs(p)
                so, do = o.costs(oparams())
                assert abs(sm - so) <= 1e-9 * so and abs(da - do) <= 1e-9 * do
                if use_rccl:  # back to RCCL on the same partition, then to the peer transport again: the state carries over
                    ps.set_option("transport", 0)
                    ps.step(p, 2 * depth + 1)
                    ps.set_option("transport", 1)
                    ps.step(p, depth + 2)
                    o.solve(oparams(), 3 * depth + 3)
                    assert np.array_equal(ps.gather_solution()[0].view(np.uint32), o.x.view(np.uint32)), "transport switched mid-stream"
                else:
                    try:
                        ps.set_option("transport", 0)
                        raise SystemExit("a communicator without RCCL accepted the RCCL transport")
                    except partition._l.FlameHipError as e:
                        assert e.code == partition._l.ERR_NORCCL
                print("rccl comm %%d: V %%d, %%d parts, depth %%d: %%d exchanges through the peer transport, resident tiles %%d, bit-exact" %% (
                    use_rccl, V, k, depth, ps.info("exchanges"), int(ps.info("persist_launches", 0) > 0)))
print("peer transport ok")
''' % ROOT

# two PROCESSES on the one GPU, no RCCL between them: inbox handles travel through files (hipIpcGetMemHandle /
# hipIpcOpenMemHandle), every exchange is a push into the OTHER process' uncached inbox and a bounded wait for its flags
RANK = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, %r)
from flame_ros_amd import graphgen, partition
from flame_ros_amd.regularizer import default_params
rank, world, tmp = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
V, depth, iters = 20000, 8, 61
g = graphgen.synthetic(V, seed=3)
p = default_params()
with partition.Communicator(0, rank, world, None) as comm:
    with partition.Partition(comm, g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, parts_per_rank=2, halo_depth=depth) as ps:
        open(os.path.join(tmp, "blob%%d.tmp" %% rank), "wb").write(ps.peer_blob())
        os.rename(os.path.join(tmp, "blob%%d.tmp" %% rank), os.path.join(tmp, "blob%%d" %% rank))
        blobs = b""
        for r in range(world):
            f = os.path.join(tmp, "blob%%d" %% r)
            t0 = time.time()
            while not os.path.exists(f):
                assert time.time() - t0 < 120, "rank %%d never published its inbox" %% r
                time.sleep(0.01)
            blobs += open(f, "rb").read()
        ps.peer_connect(blobs)
        ps.set_option("transport", 1)
        ps.step(p, iters // 3)
        ps.step(p, iters - iters // 3)
        x, w1, w2, q = ps.gather_solution()   # (no RCCL: what THIS rank owns, zero elsewhere)
        assert ps.info("peer_timeouts") == 0 and ps.info("exchanges") == (iters - 1) // depth
        np.savez(os.path.join(tmp, "out%%d.npz" %% rank), x=x, w1=w1, w2=w2, q=q, part=ps.array("part"))
        # keep the inbox alive until every rank has finished reading / writing
        open(os.path.join(tmp, "done%%d" %% rank), "w").write("1")
        t0 = time.time()
        while not all(os.path.exists(os.path.join(tmp, "done%%d" %% r)) for r in range(world)):
            assert time.time() - t0 < 120
            time.sleep(0.01)
print("rank %%d ok" %% rank)
''' % ROOT


def test_peer_transport_world_1(gpu):
    out = subprocess.run([sys.executable, "-c", WORLD1], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "peer transport ok" in out.stdout, out.stdout[-3000:] + out.stderr[-5000:]


def test_peer_transport_between_two_processes_on_one_gpu(gpu, tmp_path):
    """hipIpc-mapped inboxes: rank 0 and rank 1 are separate processes (two parts each), no RCCL, no torch -- the gathered
    solution, rank by rank, is the oracle's bit for bit."""
    import numpy as np
    from flame_ros_amd import graphgen
    from oracle import COracle
    from oracle.cbind import default_params as oparams
    world = 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", RANK, str(r), str(world), str(tmp_path)], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    for pr in procs:
        try:
            outs.append(pr.communicate(timeout=600))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (pr, (so, se)) in enumerate(zip(procs, outs)):
        assert pr.returncode == 0 and ("rank %d ok" % r) in so, (r, so[-2000:], se[-4000:])
    g = graphgen.synthetic(20000, seed=3)
    o = COracle(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt)
    o.solve(oparams(), 61)
    x = np.zeros(g.V, np.uint32)
    q = np.zeros((g.E, 3), np.uint32)
    for r in range(world):
        d = np.load(str(tmp_path / ("out%d.npz" % r)))
        x |= d["x"].view(np.uint32)   # (every vertex / edge is owned by exactly one rank, the others hold zero bits)
        q |= d["q"].view(np.uint32)
    assert np.array_equal(x, o.x.view(np.uint32)), "x"
    assert np.array_equal(q, np.asarray(o.q, np.float32).view(np.uint32)), "q"
