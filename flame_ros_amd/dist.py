"""Multi-GPU modes of the regulariser (one process per GPU, torch.distributed; "nccl" = RCCL).

SURVEY.md 8(e).  The reference is a single CPU process (no NCCL/MPI anywhere), so there is no
call pattern to mirror; two modes exist because the path shards in two ways:

* replicas  -- independent frames (graphs) per rank, no data-path collective: the natural
               data-parallel axis of FLaME (one depth-graph per camera frame).  Weak scaling.
* partition -- ONE graph cut into `world` subdomains by recursive coordinate bisection; each rank
               holds its own vertices plus D halo rings and iterates D times between neighbour
               exchanges of the full solver state (exact: each iteration invalidates one ring).
               The exchange is `batch_isend_irecv` (ncclSend/ncclRecv over xGMI) on device buffers
               filled by the halo pack/unpack kernels of libflame_hip.so.  METIS is not available in
               this image, RCB is used instead (planar graph: cut ~ sqrt(V)).

Because every vertex sums its incident edges in ascending ORIGINAL edge id on every path, the
partitioned result is bit-identical to the single-GPU (and oracle) result.

The local solver is pluggable (`SubdomainSolver` protocol) so the world_size-2 gloo tests can run
the exchange logic on CPU with the oracle as the local solver; the product class is
`HipSubdomainSolver` (GPU only, no fallback).
"""
from dataclasses import dataclass

import numpy as np

# halo records (float32 words): what changes per exchange -- {x, w1, w2, xb, w1b, w2b} per vertex,
# {q1, q2, q3} per edge (include/flame_hip.h, flame_hip_halo_pack)
VREC, EREC = 6, 3


# ---------------------------------------------------------------- replicas mode
def shard_frames(num_frames, rank, world):
    """Frames (independent graphs) handled by `rank`: round-robin, no communication."""
    return list(range(rank, num_frames, world))


# ---------------------------------------------------------------- partition mode
def rcb_parts(pos, nparts):
    """Recursive coordinate bisection into `nparts` near-equal parts; returns part id per vertex."""
    part = np.zeros(len(pos), np.int32)

    def rec(idx, lo, n):
        if n == 1 or len(idx) <= 1:
            part[idx] = lo
            return
        ext = pos[idx].max(0) - pos[idx].min(0)
        axis = int(ext[1] > ext[0])
        n1 = n // 2
        k = (len(idx) * n1) // n
        order = idx[np.lexsort((idx, pos[idx, axis]))]
        rec(order[:k], lo, n1)
        rec(order[k:], lo + n1, n - n1)

    rec(np.arange(len(pos)), 0, nparts)
    return part


@dataclass
class Subdomain:
    rank: int
    depth: int
    vid: np.ndarray        # [n_ext] global vertex ids, own first (ascending), then halo (ascending)
    n_own: int
    ring: np.ndarray       # [n_ext] graph distance from the own set
    eid: np.ndarray        # [e_loc] global edge ids, ascending (keeps every vertex's sum order)
    edges: np.ndarray      # [e_loc,2] local vertex ids, orientation preserved
    e_owned: np.ndarray    # [e_loc] bool: this rank owns the edge (= owns its source vertex)
    recv_v: dict           # owner rank -> local halo vertex ids whose state that rank sends
    recv_e: dict           # owner rank -> local edge ids whose q that rank sends


def build_subdomain(pos, edges, part, rank, depth):
    """Own vertices of `rank` + `depth` halo rings, the local edge set, and what to receive."""
    V, E = len(pos), len(edges)
    own = np.flatnonzero(part == rank)
    # level-synchronous breadth-first rings from the own set over a CSR of the undirected graph:
    # only the `depth` rings around this rank's vertices are ever touched
    ends = np.concatenate([edges[:, 0], edges[:, 1]]).astype(np.int64)
    nbrs = np.concatenate([edges[:, 1], edges[:, 0]]).astype(np.int64)
    order = np.argsort(ends, kind="stable")
    indices = nbrs[order]
    indptr = np.zeros(V + 1, np.int64)
    np.cumsum(np.bincount(ends, minlength=V), out=indptr[1:])
    dist = np.full(V, np.inf)
    dist[own] = 0
    frontier = own
    for r in range(1, depth + 1):
        if len(frontier) == 0:
            break
        starts, stops = indptr[frontier], indptr[frontier + 1]
        cnt = stops - starts
        # concatenated neighbour lists of the frontier
        idx = np.repeat(starts - np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt) + np.arange(int(cnt.sum()))
        cand = np.unique(indices[idx])
        frontier = cand[np.isinf(dist[cand])]
        dist[frontier] = r
    halo = np.flatnonzero((dist > 0) & (dist <= depth))
    vid = np.concatenate([own, halo]).astype(np.int64)
    ring_g = np.full(V, depth + 1, np.int64)
    ring_g[vid] = dist[vid].astype(np.int64)
    lid = -np.ones(V, np.int64)
    lid[vid] = np.arange(len(vid))
    ri, rj = ring_g[edges[:, 0]], ring_g[edges[:, 1]]
    keep = (ri <= depth) & (rj <= depth) & (np.minimum(ri, rj) < max(depth, 1))
    eid = np.flatnonzero(keep)
    loc = lid[edges[eid]].astype(np.int32)
    e_owner = part[edges[eid, 0]]
    recv_v, recv_e = {}, {}
    for r in np.unique(part[halo]) if len(halo) else []:
        recv_v[int(r)] = (len(own) + np.flatnonzero(part[halo] == r)).astype(np.int32)
    for r in np.unique(e_owner):
        if r != rank:
            recv_e[int(r)] = np.flatnonzero(e_owner == r).astype(np.int32)
    return Subdomain(rank, depth, vid, len(own), ring_g[vid], eid, loc, e_owner == rank, recv_v, recv_e)


class PartitionedSolver:
    """Drives `num_iters` PD iterations of ONE graph over all ranks of the default process group.

    make_solver(sub, pos, edges, alpha, beta, z, wgt, x0) must return an object with
      halo_register(send_v, send_e, recv_v, recv_e); halo_pack() -> tensor; halo_unpack(tensor);
      step(params, n); download() -> (x, w1, w2, q)
    operating on the subdomain's LOCAL numbering.
    """

    def __init__(self, pos, edges, alpha, beta, z, wgt, make_solver, depth=8, x0=None):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 2)
        edges = np.ascontiguousarray(edges, np.int32).reshape(-1, 2)
        self.V, self.E, self.depth = len(pos), len(edges), depth
        self.part = rcb_parts(pos, self.world)
        sub = self.sub = build_subdomain(pos, edges, self.part, self.rank, depth)
        f = lambda a: np.ascontiguousarray(a, np.float32)  # noqa: E731
        self.solver = make_solver(sub, pos[sub.vid], sub.edges, f(alpha)[sub.eid], f(beta)[sub.eid],
                                  f(z)[sub.vid], f(wgt)[sub.vid],
                                  None if x0 is None else f(x0)[sub.vid])
        # tell every owner which of its vertices / edges (GLOBAL ids, int64 arrays) this rank needs;
        # the owner maps them to its local numbering with a binary search (own vertices and the local
        # edge list are ascending in global id): no per-element Python anywhere
        none = np.zeros(0, np.int32)
        want = {int(r): (sub.vid[sub.recv_v.get(r, none)], sub.eid[sub.recv_e.get(r, none)])
                for r in set(sub.recv_v) | set(sub.recv_e)}
        allwant = [None] * self.world
        dist.all_gather_object(allwant, want)
        own_gid = sub.vid[:sub.n_own]
        self.peers = sorted(set(want) | {r for r in range(self.world) if self.rank in allwant[r]})
        send_v, send_e, recv_v, recv_e = [], [], [], []
        self.send_cnt, self.recv_cnt = {}, {}
        for r in self.peers:
            wv, we = allwant[r].get(self.rank, (np.zeros(0, np.int64), np.zeros(0, np.int64)))
            sv = np.searchsorted(own_gid, wv).astype(np.int32)
            se = np.searchsorted(sub.eid, we).astype(np.int32)
            assert np.array_equal(own_gid[sv], wv) and np.array_equal(sub.eid[se], we), "request for state this rank does not own"
            rv, re_ = sub.recv_v.get(r, none), sub.recv_e.get(r, none)
            self.send_cnt[r] = (len(sv), len(se))
            self.recv_cnt[r] = (len(rv), len(re_))
            send_v.append(sv); send_e.append(se); recv_v.append(rv); recv_e.append(re_)
        cat = lambda a: np.concatenate(a).astype(np.int32) if a else none  # noqa: E731
        send_v, send_e, recv_v, recv_e = cat(send_v), cat(send_e), cat(recv_v), cat(recv_e)
        self.n_send = (len(send_v), len(send_e))
        self.n_recv = (len(recv_v), len(recv_e))
        self.solver.halo_register(send_v, send_e, recv_v, recv_e)
        self._rings_left = depth  # a fresh upload holds exact state on every ring
        self._sbuf = self._rbuf = None  # persistent exchange buffers, allocated on the first exchange
        self._ops_cache = None

    # packed buffer layout: all vertex records (VREC floats each, peers in order) then all edge
    # records (EREC floats each, peers in order) -> per-peer messages are two slices each
    def _slices(self, cnt, n):
        out, ov, oe = {}, 0, VREC * n[0]
        for r in self.peers:
            nv, ne = cnt[r]
            out[r] = ((ov, ov + VREC * nv), (oe, oe + EREC * ne))
            ov += VREC * nv
            oe += EREC * ne
        return out

    def exchange(self):
        if self.world == 1 or not self.peers:
            return
        ctx = getattr(self.solver, "stream_context", None)
        if ctx is None:
            return self._exchange()
        with ctx():  # pack, P2P and unpack are all ordered on the solver's stream
            return self._exchange()

    def _exchange(self):
        """pack -> P2P -> unpack.  The send and receive buffers are allocated ONCE (first exchange) and
        the P2POp list is built once; with the nccl backend everything is enqueued on the solver's
        stream (Work.wait() of an NCCL op orders the stream, it does not block the host), so a step()
        returns without a host synchronisation -- only download() / costs() synchronise."""
        dist = self.dist
        if self._sbuf is None:
            first = self.solver.halo_pack()  # (a solver without persistent buffers returns a new one)
            # gloo has no device-to-device path: stage through the host (tests with several ranks on
            # ONE GPU; the product backend is nccl = RCCL, device buffers straight into ncclSend/Recv)
            self._staged = first.is_cuda and dist.get_backend() == "gloo"
            self._dev = first.device
            self._sbuf = first.cpu() if self._staged else first
            self._rbuf = self._sbuf.new_empty(VREC * self.n_recv[0] + EREC * self.n_recv[1])
            self._rdev = self._rbuf.to(self._dev) if self._staged else self._rbuf
            ops, ssl, rsl = [], self._slices(self.send_cnt, self.n_send), self._slices(self.recv_cnt, self.n_recv)
            for r in self.peers:
                for a, b in ssl[r]:
                    if b > a:
                        ops.append(dist.P2POp(dist.isend, self._sbuf[a:b], r))
                for a, b in rsl[r]:
                    if b > a:
                        ops.append(dist.P2POp(dist.irecv, self._rbuf[a:b], r))
            self._ops_cache = ops
            self._pack_into = getattr(self.solver, "halo_pack_into", None)
        elif self._pack_into is not None and not self._staged:
            self._pack_into(self._sbuf)  # in place: no allocation
        else:
            packed = self.solver.halo_pack()
            self._sbuf.copy_(packed)  # (gloo test path / solvers without halo_pack_into)
        if self._ops_cache:
            for w in dist.batch_isend_irecv(self._ops_cache):
                w.wait()
        if self._staged:
            self._rdev.copy_(self._rbuf)
        self.solver.halo_unpack(self._rdev)

    def step(self, params, num_iters):
        """Every local iteration invalidates one halo ring: `_rings_left` counts how many more
        iterations the halo still supports; an exchange (exact state from the owners) resets it to
        the halo depth.  Successive step() calls continue where the previous one stopped."""
        done = 0
        while done < num_iters:
            if self._rings_left == 0:
                self.exchange()
                self._rings_left = self.depth
            n = min(self._rings_left, num_iters - done)
            self.solver.step(params, n)
            self._rings_left -= n
            done += n

    def costs(self, params):
        """nltgv2_total_smoothness_cost / nltgv2_total_data_cost of the WHOLE graph (the stat keys read
        at reference src/utils.cc:131-136): every rank sums the edges and vertices it owns, one
        all-reduce of 2 doubles (SURVEY.md 8e "final cost reduction")."""
        import torch
        s = self.sub
        if self._rings_left == 0 and self.depth > 0:  # an owned edge reads its target in ring 1
            self.exchange()
            self._rings_left = self.depth
        vmask = np.zeros(len(s.vid), np.uint8)
        vmask[:s.n_own] = 1
        sm, da = self.solver.costs_owned(params, vmask, s.e_owned.astype(np.uint8))
        t = torch.tensor([sm, da], dtype=torch.float64)
        if self.world > 1:
            if self.dist.get_backend() == "nccl":
                t = t.cuda()
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t[0]), float(t[1])

    def gather_solution(self):
        """x, w1, w2 (V) and q (E,3) of the whole graph on every rank (verification helper)."""
        x, w1, w2, q = self.solver.download()
        s = self.sub
        own_e = np.flatnonzero(s.e_owned)
        mine = (s.vid[:s.n_own], x[:s.n_own], w1[:s.n_own], w2[:s.n_own], s.eid[own_e], q[own_e])
        parts = [None] * self.world
        self.dist.all_gather_object(parts, mine)
        X, W1, W2 = (np.zeros(self.V, np.float32) for _ in range(3))
        Q = np.zeros((self.E, 3), np.float32)
        for vid, px, p1, p2, eid, pq in parts:
            X[vid], W1[vid], W2[vid] = px, p1, p2
            Q[eid] = pq
        return X, W1, W2, Q


class HipSubdomainSolver:
    """The product local solver: one subdomain on one MI355X through libflame_hip.so.  Halo
    buffers are torch CUDA tensors; kernels and RCCL ops are ordered on torch's current stream."""

    def __init__(self, sub, pos, edges, alpha, beta, z, wgt, x0, device=0, stream=None, **options):
        import ctypes as C

        import torch

        from . import lib as _l
        from .regularizer import GraphRegularizer
        self._C, self._l, self.torch = C, _l, torch
        self.device = torch.device("cuda", device)
        # One explicit (non-default) torch stream orders everything: solver kernels, halo
        # pack/unpack, torch copies and the RCCL P2P ops issued under stream_context().  (A NULL
        # stream would mean "the handle's own stream" to the C ABI, unordered with torch.)
        self.stream = stream if stream is not None else torch.cuda.Stream(self.device)
        # the D-iteration local solve is captured once into a hipGraph and replayed on this stream
        self.reg = GraphRegularizer(pos, edges, alpha, beta, z, wgt, x0=x0, device=device, **options)
        self.n_send = self.n_recv = (0, 0)

    def stream_context(self):
        return self.torch.cuda.stream(self.stream)

    def _stream(self):
        return self._C.c_void_p(self.stream.cuda_stream)

    def halo_register(self, send_v, send_e, recv_v, recv_e):
        p = lambda a: a.ctypes.data_as(self._C.c_void_p)  # noqa: E731
        self._l.check(self.reg._lib.flame_hip_halo_register(
            self.reg._h, len(send_v), p(send_v), len(send_e), p(send_e), len(recv_v), p(recv_v),
            len(recv_e), p(recv_e)), "flame_hip_halo_register")
        self.n_send, self.n_recv = (len(send_v), len(send_e)), (len(recv_v), len(recv_e))

    def halo_pack(self):
        with self.stream_context():
            buf = self.torch.empty(VREC * self.n_send[0] + EREC * self.n_send[1],
                                   dtype=self.torch.float32, device=self.device)
        return self.halo_pack_into(buf)

    def halo_pack_into(self, buf):
        """Pack into a caller-owned (persistent) device buffer: no allocation per exchange."""
        assert buf.is_cuda and buf.dtype == self.torch.float32 and buf.is_contiguous()
        self._l.check(self.reg._lib.flame_hip_halo_pack(self.reg._h, self._C.c_void_p(buf.data_ptr()),
                                                        self._stream()), "flame_hip_halo_pack")
        return buf

    def costs_owned(self, params, vmask, emask):
        self.stream.synchronize()
        return self.reg.costs_masked(params, vmask, emask)

    def halo_unpack(self, buf):
        assert buf.is_cuda and buf.dtype == self.torch.float32 and buf.is_contiguous()
        self._l.check(self.reg._lib.flame_hip_halo_unpack(self.reg._h, self._C.c_void_p(buf.data_ptr()),
                                                          self._stream()), "flame_hip_halo_unpack")

    def step(self, params, n):
        self.reg.step(params, n, stream=self._stream(), sync=False)

    def download(self):
        self.stream.synchronize()
        return self.reg.download()


def make_hip_solver(device=0, **options):
    def f(sub, pos, edges, alpha, beta, z, wgt, x0):
        return HipSubdomainSolver(sub, pos, edges, alpha, beta, z, wgt, x0, device=device, **options)
    return f
