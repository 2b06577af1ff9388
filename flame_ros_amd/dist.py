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

    parts_per_rank = k > 1 over-decomposes: the graph is cut into world * k subdomains and a rank holds k
    consecutive ones (part p lives on rank p // k); halo records between two parts of one rank travel the
    same way as all others -- a send / receive pair of the rank with ITSELF inside the one batch of P2P
    operations (ncclGroupStart .. ncclSend / ncclRecv to the own rank .. ncclGroupEnd).  world 1, k 2 is
    how the RCCL path is exercised on a single GPU (tests/test_gpu_nccl_self.py).
    """

    def __init__(self, pos, edges, alpha, beta, z, wgt, make_solver, depth=8, x0=None, parts_per_rank=1):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        k = self.k = int(parts_per_rank)
        nparts = self.nparts = self.world * k
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 2)
        edges = np.ascontiguousarray(edges, np.int32).reshape(-1, 2)
        self.V, self.E, self.depth = len(pos), len(edges), depth
        self.part = rcb_parts(pos, nparts)
        self.my_parts = list(range(self.rank * k, (self.rank + 1) * k))
        f = lambda a: np.ascontiguousarray(a, np.float32)  # noqa: E731
        self.subs, self.solvers = [], []
        for p in self.my_parts:
            sub = build_subdomain(pos, edges, self.part, p, depth)
            self.subs.append(sub)
            self.solvers.append(make_solver(sub, pos[sub.vid], sub.edges, f(alpha)[sub.eid], f(beta)[sub.eid],
                                            f(z)[sub.vid], f(wgt)[sub.vid],
                                            None if x0 is None else f(x0)[sub.vid]))
        self.sub, self.solver = self.subs[0], self.solvers[0]  # (the one-part-per-rank names)
        # tell every owner which of its vertices / edges (GLOBAL ids, int64 arrays) a part needs; the owner
        # maps them to its local numbering with a binary search (own vertices and the local edge list are
        # ascending in global id): no per-element Python anywhere
        none = np.zeros(0, np.int32)
        want = {}  # (owner part, wanting part) -> (global vertex ids, global edge ids)
        for sub in self.subs:
            for o in set(sub.recv_v) | set(sub.recv_e):
                want[(int(o), sub.rank)] = (sub.vid[sub.recv_v.get(o, none)], sub.eid[sub.recv_e.get(o, none)])
        allwant = [None] * self.world
        dist.all_gather_object(allwant, want)
        wants = {}
        for w in allwant:
            wants.update(w)
        # per local part: its peer parts in ascending order, send / receive lists and counts per peer
        self.peers_of, self.send_cnt, self.recv_cnt, self.n_send_of, self.n_recv_of = [], [], [], [], []
        for sub, solver in zip(self.subs, self.solvers):
            me = sub.rank
            peers = sorted({w for (o, w) in wants if o == me} | {o for (o, w) in wants if w == me})
            own_gid = sub.vid[:sub.n_own]
            send_v, send_e, recv_v, recv_e, sc, rc = [], [], [], [], {}, {}
            for r in peers:
                wv, we = wants.get((me, r), (np.zeros(0, np.int64), np.zeros(0, np.int64)))
                sv = np.searchsorted(own_gid, wv).astype(np.int32)
                se = np.searchsorted(sub.eid, we).astype(np.int32)
                assert np.array_equal(own_gid[sv], wv) and np.array_equal(sub.eid[se], we), "request for state this part does not own"
                rv, re_ = sub.recv_v.get(r, none), sub.recv_e.get(r, none)
                sc[r] = (len(sv), len(se))
                rc[r] = (len(rv), len(re_))
                send_v.append(sv); send_e.append(se); recv_v.append(rv); recv_e.append(re_)
            cat = lambda a: np.concatenate(a).astype(np.int32) if a else none  # noqa: E731
            send_v, send_e, recv_v, recv_e = cat(send_v), cat(send_e), cat(recv_v), cat(recv_e)
            solver.halo_register(send_v, send_e, recv_v, recv_e)
            self.peers_of.append(peers); self.send_cnt.append(sc); self.recv_cnt.append(rc)
            self.n_send_of.append((len(send_v), len(send_e))); self.n_recv_of.append((len(recv_v), len(recv_e)))
        self.peers, self.n_send, self.n_recv = self.peers_of[0], self.n_send_of[0], self.n_recv_of[0]
        self._rings_left = depth  # a fresh upload holds exact state on every ring
        self._sbuf = self._rbuf = None  # persistent exchange buffers, allocated on the first exchange
        self._ops_cache = None

    # packed buffer layout of one part: all vertex records (VREC floats each, peers in order) then all
    # edge records (EREC floats each, peers in order) -> per-peer messages are two slices each
    @staticmethod
    def _slices(peers, cnt, n):
        out, ov, oe = {}, 0, VREC * n[0]
        for r in peers:
            nv, ne = cnt[r]
            out[r] = ((ov, ov + VREC * nv), (oe, oe + EREC * ne))
            ov += VREC * nv
            oe += EREC * ne
        return out

    def exchange(self):
        if self.nparts == 1 or not any(self.peers_of):
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
        returns without a host synchronisation -- only download() / costs() synchronise.
        Messages between one pair of ranks match by ORDER (no tags in ncclSend / ncclRecv): both sides
        issue them sorted by (sending part, receiving part, vertex records before edge records)."""
        dist = self.dist
        if self._sbuf is None:
            firsts = [sv.halo_pack() for sv in self.solvers]  # (a solver without persistent buffers returns a new one)
            # gloo has no device-to-device path: stage through the host (tests with several ranks on
            # ONE GPU; the product backend is nccl = RCCL, device buffers straight into ncclSend/Recv)
            self._staged = firsts[0].is_cuda and dist.get_backend() == "gloo"
            self._dev = firsts[0].device
            self._sbuf = [f.cpu() if self._staged else f for f in firsts]
            self._rbuf = [self._sbuf[i].new_empty(VREC * n[0] + EREC * n[1]) for i, n in enumerate(self.n_recv_of)]
            self._rdev = [r.to(self._dev) if self._staged else r for r in self._rbuf]
            sends, recvs = [], []
            for i, me in enumerate(self.my_parts):
                ssl = self._slices(self.peers_of[i], self.send_cnt[i], self.n_send_of[i])
                rsl = self._slices(self.peers_of[i], self.recv_cnt[i], self.n_recv_of[i])
                for r in self.peers_of[i]:
                    for kind, (a, b) in enumerate(ssl[r]):
                        if b > a:
                            sends.append(((me, r, kind), dist.P2POp(dist.isend, self._sbuf[i][a:b], r // self.k)))
                    for kind, (a, b) in enumerate(rsl[r]):
                        if b > a:
                            recvs.append(((r, me, kind), dist.P2POp(dist.irecv, self._rbuf[i][a:b], r // self.k)))
            self._self_copies = []
            if dist.get_backend() != "nccl":  # gloo has no pair of a rank with itself: those records are copied
                rmap = {key: op for key, op in recvs if op.peer == self.rank}
                self._self_copies = [(rmap[key].tensor, op.tensor) for key, op in sends if op.peer == self.rank]
                sends = [t for t in sends if t[1].peer != self.rank]
                recvs = [t for t in recvs if t[1].peer != self.rank]
            self._ops_cache = [op for _, op in sorted(sends, key=lambda t: t[0])] + \
                              [op for _, op in sorted(recvs, key=lambda t: t[0])]
            self._pack_into = [getattr(sv, "halo_pack_into", None) for sv in self.solvers]
        else:
            for i, sv in enumerate(self.solvers):
                if self._pack_into[i] is not None and not self._staged:
                    self._pack_into[i](self._sbuf[i])  # in place: no allocation
                else:
                    self._sbuf[i].copy_(sv.halo_pack())  # (gloo test path / solvers without halo_pack_into)
        if self._ops_cache:
            for w in dist.batch_isend_irecv(self._ops_cache):
                w.wait()
        for dst, src in self._self_copies:
            dst.copy_(src)
        for i, sv in enumerate(self.solvers):
            if self._staged:
                self._rdev[i].copy_(self._rbuf[i])
            sv.halo_unpack(self._rdev[i])

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
            for sv in self.solvers:
                sv.step(params, n)
            self._rings_left -= n
            done += n

    def costs(self, params):
        """nltgv2_total_smoothness_cost / nltgv2_total_data_cost of the WHOLE graph (the stat keys read
        at reference src/utils.cc:131-136): every part sums the edges and vertices it owns, one
        all-reduce of 2 doubles (SURVEY.md 8e "final cost reduction"; with nccl a device tensor, also at
        world 1, so that the collective itself runs)."""
        import torch
        if self._rings_left == 0 and self.depth > 0:  # an owned edge reads its target in ring 1
            self.exchange()
            self._rings_left = self.depth
        sm = da = 0.0
        for s, sv in zip(self.subs, self.solvers):
            vmask = np.zeros(len(s.vid), np.uint8)
            vmask[:s.n_own] = 1
            a, b = sv.costs_owned(params, vmask, s.e_owned.astype(np.uint8))
            sm += a
            da += b
        t = torch.tensor([sm, da], dtype=torch.float64)
        nccl = self.dist.get_backend() == "nccl"
        if self.world > 1 or nccl:
            if nccl:
                t = t.cuda()
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t[0]), float(t[1])

    def gather_solution(self):
        """x, w1, w2 (V) and q (E,3) of the whole graph on every rank (verification helper)."""
        mine = []
        for s, sv in zip(self.subs, self.solvers):
            x, w1, w2, q = sv.download()
            own_e = np.flatnonzero(s.e_owned)
            mine.append((s.vid[:s.n_own], x[:s.n_own], w1[:s.n_own], w2[:s.n_own], s.eid[own_e], q[own_e]))
        parts = [None] * self.world
        self.dist.all_gather_object(parts, mine)
        X, W1, W2 = (np.zeros(self.V, np.float32) for _ in range(3))
        Q = np.zeros((self.E, 3), np.float32)
        for plist in parts:
            for vid, px, p1, p2, eid, pq in plist:
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
        # resident tiles assume the whole chip: off when several ranks share one GPU (the gloo development
        # check of the N>1 paths), where the ranks' launches would starve each other until they time out
        try:
            import torch.distributed as tdist
            if tdist.is_initialized() and tdist.get_world_size() > torch.cuda.device_count():
                options.setdefault("persist", 0)
        except Exception:  # noqa: BLE001
            pass
        # the D-iteration local solve runs as one launch of resident tiles (or a replayed hipGraph) on this stream
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
    """Factory for PartitionedSolver: every subdomain this rank holds (parts_per_rank of them) is solved on
    ONE torch stream, which also orders the packs, the P2P operations and the unpacks between them."""
    shared = {}

    def f(sub, pos, edges, alpha, beta, z, wgt, x0):
        if "stream" not in shared:
            import torch
            shared["stream"] = torch.cuda.Stream(torch.device("cuda", device))
        return HipSubdomainSolver(sub, pos, edges, alpha, beta, z, wgt, x0, device=device, stream=shared["stream"], **options)
    return f
