"""-m gpu: the RCCL code of the partition mode executed on the ONE GPU of the test box.

`torch.distributed` backend "nccl" (= RCCL on ROCm) initialised with world size 1 and `device_id=`; the graph is cut
into parts_per_rank subdomains that rank 0 all holds, so every halo record travels through
`batch_isend_irecv` = ncclGroupStart / ncclSend + ncclRecv to the OWN rank / ncclGroupEnd on device buffers that
`k_halo_pack` filled on the solver's stream, is unpacked by `k_halo_unpack` behind `Work.wait()` on the same stream,
and the cost reduction is an `all_reduce` of a device tensor.  Bit-exact against the oracle.  What this proves before
a multi-GPU node ever runs it: RCCL loads and initialises beside libflame_hip.so, the stream ordering of
pack -> P2P -> unpack -> next solve holds without a host synchronisation, message matching by order is right.
(Runs in a process of its own: a process group cannot be re-initialised inside the pytest process.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import os, socket, sys
import numpy as np
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from flame_ros_amd import dist as fdist, graphgen
from flame_ros_amd.regularizer import default_params
from oracle import COracle
from oracle.cbind import default_params as oparams
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
for V, k, depth, iters in ((6000, 2, 8, 50), (9000, 3, 4, 23)):
    g = graphgen.synthetic(V, seed=5)
    ps = fdist.PartitionedSolver(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, fdist.make_hip_solver(0), depth=depth,
                                 parts_per_rank=k)
    assert len(ps.subs) == k and all(len(p) >= 1 for p in ps.peers_of)
    p = default_params()
    ps.step(p, iters // 2)            # (no host synchronisation inside: pack, P2P, unpack and the solves are stream-ordered)
    ps.step(p, iters - iters // 2)
    assert ps._ops_cache and not ps._staged and ps._sbuf[0].is_cuda and ps._rbuf[0].is_cuda
    x, w1, w2, q = ps.gather_solution()
    sm, da = ps.costs(p)              # all_reduce of a CUDA tensor over RCCL
    o = COracle(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt)
    o.solve(oparams(), iters)
    for name, got, want in (("x", x, o.x), ("w1", w1, o.w1), ("w2", w2, o.w2), ("q", q, o.q)):
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (V, k, name)
    so, do = o.costs(oparams())
    assert abs(sm - so) <= 1e-9 * so and abs(da - do) <= 1e-9 * do, (sm, so, da, do)
    n_ops = len(ps._ops_cache)
    print("V %%d, %%d parts on rank 0, depth %%d: %%d P2P ops per exchange, bit-exact" %% (V, k, depth, n_ops))
dist.barrier()
dist.destroy_process_group()
print("nccl self ok")
''' % ROOT


def test_rccl_halo_exchange_with_itself_and_cost_allreduce(gpu):
    env = dict(os.environ)
    out = subprocess.run([sys.executable, "-c", CODE], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "nccl self ok" in out.stdout, out.stdout[-3000:] + out.stderr[-5000:]
