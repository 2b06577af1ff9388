"""-m gpu: the N>1 paths run by REAL processes on the one GPU of the test box (gloo between the
ranks, halo buffers staged through the host): `PartitionedSolver` + `HipSubdomainSolver` (product
classes) for 2 and 3 ranks, bit-identical to the oracle, and `bench.py` itself under
`torch.distributed.run` in both N>1 modes (replicas, partition).  On a multi-GPU node the same code
runs over nccl = RCCL with device buffers."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from oracle import COracle
from oracle.cbind import default_params as oracle_params
from tests.util import graphgen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, V, depth, iters, out):
    import torch
    import torch.distributed as dist
    from flame_ros_amd import dist as fdist
    from flame_ros_amd.regularizer import default_params
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        g = graphgen.synthetic(V, seed=31)
        ps = fdist.PartitionedSolver(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, fdist.make_hip_solver(0),
                                     depth=depth)
        ps.step(default_params(), iters // 2)          # two calls: the second continues on the halo
        ps.step(default_params(), iters - iters // 2)   # rings the first left valid
        x, w1, w2, q = ps.gather_solution()
        sm, da = ps.costs(default_params())             # owned sums (k_costs, masked) + all-reduce
        if rank == 0:
            np.savez(out, x=x, w1=w1, w2=w2, q=q, costs=np.array([sm, da]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,depth,iters", [(2, 16, 70), (3, 8, 30)])
def test_partitioned_hip_solver_processes(gpu, tmp_path, world, depth, iters):
    V = 9000
    out = str(tmp_path / "res.npz")
    mp.spawn(_worker, args=(world, _free_port(), V, depth, iters, out), nprocs=world, join=True)
    g = graphgen.synthetic(V, seed=31)
    o = COracle(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt)
    o.solve(oracle_params(), iters)
    r = np.load(out)
    for k, want in (("x", o.x), ("w1", o.w1), ("w2", o.w2), ("q", o.q)):
        assert np.array_equal(r[k].view(np.uint32), want.view(np.uint32)), k
    so, do = o.costs(oracle_params())
    assert abs(r["costs"][0] - so) <= 1e-9 * so and abs(r["costs"][1] - do) <= 1e-9 * do, (r["costs"], so, do)


@pytest.mark.parametrize("mode", ["replicas", "partition"])
def test_bench_runs_with_two_ranks(gpu, mode):
    env = dict(os.environ, FLAME_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps",
           "3", "--warmup", "1", "--workload", "5k", "--mode", mode, "--halo-depth", "8"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["steps"] == 3
    assert d["scaling"] == ("strong" if mode == "partition" else "weak")
    if mode == "partition":
        assert d["partition_torch_harness"]["send_bytes_rank0"] > 0 and d["partition_torch_harness"]["exchange_us"] > 0
        assert d["config"]["parallelism"].startswith("partition2")
    else:
        assert d["config"]["parallelism"] == "replicas2"


def test_bench_starts_its_own_ranks(gpu):
    """`python bench.py --gpus 2` with no launcher in the command: bench.py starts the two ranks itself and the line
    says n_gpus 2 (gloo development backend: both ranks on the one GPU of this box); with the product backend and
    fewer GPUs than ranks it refuses loudly instead of quietly running one rank."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "5k"]
    p = subprocess.run(cmd, cwd=ROOT, env=dict(env, FLAME_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "replicas2" and d["value"] > 0
    if torch.cuda.device_count() < 2:
        p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode != 0 and "only 1 GPU(s) visible" in (p.stdout + p.stderr), p.stdout[-1000:] + p.stderr[-2000:]


LIBPART = r"""
import json, sys
sys.path.insert(0, %r)
import bench
from flame_ros_amd import partition
import torch
d = bench.library_partition(0, 1, 0, partition.unique_id(), torch.cuda.synchronize, lambda v: v, parts_per_rank=PARTS, steps=3)
print("LIBPART " + json.dumps(d))
"""


@pytest.mark.parametrize("parts", [2, 8])
def test_bench_library_partition_block_world_1(gpu, parts):
    """the `partition` block `bench.py --gpus N` prints at N > 1 is measured through flame_hip_comm_* /
    flame_hip_part_* (not the torch harness).  The same function on the one GPU there is: world 1 x 2 parts (the 50 k graph of
    BASELINE config 4) and x 8 parts (the 200 k graph of config 5) -- every halo record an ncclSend / ncclRecv of the rank
    with itself; what the driver's first multi-GPU run will execute, minus the second GPU."""
    p = subprocess.run([sys.executable, "-c", (LIBPART % ROOT).replace("PARTS", str(parts))], cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("LIBPART ")][-1][8:])
    assert d["workload"] == ("50k" if parts == 2 else "200k") and d["rccl_ranks"] == 1 and d["bit_exact_vs_one_gpu"] is True, d
    assert d["iterations_per_s"] > 0 and d["exchange_us"] > 0 and d["exchanges_per_step"] in (d["iters_per_step"] // 16, d["iters_per_step"] // 16 + 1), d
    assert d["send_bytes_per_exchange_rank0"] > 0 and d["resident_tiles"] is True and d["solves_repeated_after_a_give_up"] == 0, d
    print(d)


def test_bench_partition_glue_at_world_1(gpu):
    """The N > 1 branch of bench.py that nobody can run here -- process group on nccl, the unique id over the torch store, the
    three library-partition variants, the watchdog, the block joined to the ONE JSON line -- forced at world 1
    (FLAME_BENCH_FORCE_PARTITION): the line must carry `partition` with RCCL's own rank count and bit-exact variants."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FLAME_BENCH_FORCE_PARTITION"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu", "--no-facade"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    pt = d["partition"]
    assert "error" not in pt and pt["rccl_ranks"] == 1 and pt["bit_exact_vs_one_gpu"] is True, pt
    assert pt["halo_depth_x2"].get("bit_exact_vs_one_gpu") is True and pt["halo_depth_x2"]["halo_depth"] == 32, pt["halo_depth_x2"]
    assert pt["two_parts_per_rank_pipelined"].get("bit_exact_vs_one_gpu") is True, pt["two_parts_per_rank_pipelined"]
    assert pt["two_parts_per_rank_pipelined"]["exchanges_pipelined"] > 0
    # r06: the same cut through the peer transport, RCCL's figures beside it
    assert pt["peer_transport"].get("bit_exact_vs_one_gpu") is True and pt["peer_transport"]["transport"] == "peer", pt["peer_transport"]
    assert pt["transport"] == "rccl"  # (one part on one rank at world 1: no exchange to time on either transport)
    assert d["n_gpus"] == 1 and d["value"] > 0
    # r06: the line's HBM bytes per launch are this run's own counters (rocprofv3 PMC child passes of bench.py) -- or, where the
    # profiler cannot count, the committed profile of the same sources: a number either way, and a plausible one
    rl = d["roofline"]
    assert rl.get("traffic") and rl.get("traffic_source"), {k: rl.get(k) for k in ("traffic", "traffic_source", "counters_source")}
    assert 0.02 < rl["measured_hbm_frac"] < 1.0 and 0.05 < rl["frac"] <= 1.0, (rl["measured_hbm_frac"], rl["frac"])
    print("traffic source:", rl["traffic_source"], "| counters:", rl.get("counters_source"))
