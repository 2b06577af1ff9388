"""CPU, world_size 2 and 3 over gloo: the N>1 paths of flame_ros_amd/dist.py.

The exchange logic (RCB partition, halo rings, request lists, packed P2P messages, D iterations
between exchanges) is the product code; the LOCAL solver is injected: here the CPU oracle (test
infrastructure), on the GPU box `HipSubdomainSolver`.  The partitioned result must be bit-identical
to the serial oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from flame_ros_amd import dist as fdist
from oracle import COracle
from oracle.cbind import default_params
from tests.util import graphgen


class OracleSubdomainSolver:
    def __init__(self, sub, pos, edges, alpha, beta, z, wgt, x0):
        self.o = COracle(pos, edges, alpha, beta, z, wgt, x0=x0)

    def halo_register(self, send_v, send_e, recv_v, recv_e):
        self.sv, self.se, self.rv, self.re = send_v, send_e, recv_v, recv_e

    def halo_pack(self):
        o = self.o
        v = np.stack([o.x, o.w1, o.w2, o.xb, o.w1b, o.w2b], 1)[self.sv]  # fdist.VREC words
        q = o.q[self.se]                                                  # fdist.EREC words
        return torch.from_numpy(np.concatenate([v.ravel(), q.ravel()]).astype(np.float32))

    def halo_unpack(self, buf):
        o, b = self.o, buf.numpy()
        nv = len(self.rv)
        v = b[:fdist.VREC * nv].reshape(nv, fdist.VREC)
        for k, a in enumerate((o.x, o.w1, o.w2, o.xb, o.w1b, o.w2b)):
            a[self.rv] = v[:, k]
        o.q[self.re] = b[fdist.VREC * nv:].reshape(-1, fdist.EREC)

    def step(self, params, n):
        self.o.solve(params, n)

    def download(self):
        return self.o.x, self.o.w1, self.o.w2, self.o.q

    def costs_owned(self, params, vmask, emask):
        # the oracle's cost terms over the owned vertices / edges (float32 terms, float64 sums)
        o = self.o
        e = np.flatnonzero(emask)
        i, j = o.edges[e, 0], o.edges[e, 1]
        d = o.pos[i] - o.pos[j]
        t = o.x[i] - o.x[j]
        t = np.float32(t - o.w1[i] * d[:, 0])  # (test-side arithmetic: compared with a tolerance)
        t = np.float32(t - o.w2[i] * d[:, 1])
        sm = (o.alpha[e] * np.abs(t)).astype(np.float64) + (o.beta[e] * np.abs(o.w1[i] - o.w1[j])).astype(np.float64) \
            + (o.beta[e] * np.abs(o.w2[i] - o.w2[j])).astype(np.float64)
        v = np.flatnonzero(vmask)
        da = ((np.float32(params.data_factor) * o.wgt[v]) * np.abs(o.x[v] - o.z[v])).astype(np.float64)
        return float(sm.sum()), float(da.sum())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, V, depth, iters, out, parts_per_rank=1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = graphgen.synthetic(V, seed=11)
        p = default_params()
        ps = fdist.PartitionedSolver(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt,
                                     lambda *a: OracleSubdomainSolver(*a), depth=depth, parts_per_rank=parts_per_rank)
        assert len(ps.subs) == parts_per_rank and ps.nparts == world * parts_per_rank
        assert ps.sub.n_own > 0 and len(ps.peers) >= 1
        # two step() calls: the second continues on whatever halo rings the first left valid
        ps.step(p, iters // 2)
        ps.step(p, iters - iters // 2)
        x, w1, w2, q = ps.gather_solution()
        sm, da = ps.costs(p)  # owned sums + one all-reduce of 2 doubles
        # replicas mode: frames are sharded round-robin, aggregate = max time over ranks
        frames = fdist.shard_frames(7, rank, world)
        t = torch.tensor([float(len(frames))])
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        assert int(t.item()) == 7
        if rank == 0:
            np.savez(out, x=x, w1=w1, w2=w2, q=q, halo=len(ps.sub.vid) - ps.sub.n_own, costs=np.array([sm, da]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,depth,iters,V", [(2, 4, 19, 1500), (3, 2, 9, 1500), (2, 1, 5, 1500),
                                                  (2, 16, 50, 6000)])  # depth 16 = the bench's default halo
def test_partitioned_solve_matches_serial_oracle(tmp_path, world, depth, iters, V):
    out = str(tmp_path / "res.npz")
    mp.spawn(_worker, args=(world, _free_port(), V, depth, iters, out), nprocs=world, join=True)
    g = graphgen.synthetic(V, seed=11)
    o = COracle(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt)
    o.solve(default_params(), iters)
    r = np.load(out)
    assert int(r["halo"]) > 0
    for k, want in (("x", o.x), ("w1", o.w1), ("w2", o.w2), ("q", o.q)):
        assert np.array_equal(r[k].view(np.uint32), want.view(np.uint32)), k
    so, do = o.costs(default_params())  # whole-graph costs = all-reduced owned sums
    assert abs(r["costs"][0] - so) <= 1e-6 * so and abs(r["costs"][1] - do) <= 1e-6 * do, (r["costs"], so, do)


@pytest.mark.parametrize("world,k,depth,iters", [(2, 2, 3, 14), (1, 3, 4, 17), (2, 3, 2, 9)])
def test_over_decomposed_partition_matches_serial_oracle(tmp_path, world, k, depth, iters):
    """parts_per_rank = k: world * k subdomains, k per rank; the records between two parts of one rank are a
    send / receive of the rank with itself inside the same batch (the form the single-GPU RCCL test uses)."""
    V = 1800
    out = str(tmp_path / "res.npz")
    mp.spawn(_worker, args=(world, _free_port(), V, depth, iters, out, k), nprocs=world, join=True)
    g = graphgen.synthetic(V, seed=11)
    o = COracle(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt)
    o.solve(default_params(), iters)
    r = np.load(out)
    for key, want in (("x", o.x), ("w1", o.w1), ("w2", o.w2), ("q", o.q)):
        assert np.array_equal(r[key].view(np.uint32), want.view(np.uint32)), key
    so, do = o.costs(default_params())
    assert abs(r["costs"][0] - so) <= 1e-6 * so and abs(r["costs"][1] - do) <= 1e-6 * do, (r["costs"], so, do)


def test_rcb_and_subdomain_structure():
    g = graphgen.synthetic(2000, seed=3)
    part = fdist.rcb_parts(g.pos, 8)
    counts = np.bincount(part, minlength=8)
    assert counts.min() >= 249 and counts.max() <= 251
    cut = int((part[g.edges[:, 0]] != part[g.edges[:, 1]]).sum())
    assert cut < 0.15 * g.E  # planar RCB cut ~ sqrt(V) per boundary
    sub = fdist.build_subdomain(g.pos, g.edges, part, 3, 3)
    assert np.all(part[sub.vid[:sub.n_own]] == 3) and np.all(sub.ring[:sub.n_own] == 0)
    assert sub.ring.max() == 3 and np.all(np.diff(sub.eid) > 0)
    assert np.array_equal(sub.vid[sub.edges], g.edges[sub.eid])  # orientation preserved
    owned = part[g.edges[sub.eid, 0]] == 3
    assert np.array_equal(owned, sub.e_owned)
    # every edge incident to an own vertex is local (so own vertices see all their neighbours)
    inc = np.isin(g.edges, sub.vid[:sub.n_own]).any(1)
    assert np.all(np.isin(np.flatnonzero(inc), sub.eid))
    assert fdist.shard_frames(10, 1, 4) == [1, 5, 9]
