"""Drives HipSubdomainSolver for `world` subdomains of ONE graph that all live on cuda:0: the halo
records travel through device buffers inside this process (what RCCL moves between GPUs at N>1).
Shared by tests/test_gpu_halo.py and tests/test_gpu_configs.py."""
import numpy as np

from flame_ros_amd import dist as fdist
from flame_ros_amd.regularizer import default_params


def run_subdomains_one_gpu(g, world, depth, iters):
    """Returns (subs, solvers) after `iters` PD iterations with an exchange every `depth`."""
    import torch
    part = fdist.rcb_parts(g.pos, world)
    subs = [fdist.build_subdomain(g.pos, g.edges, part, r, depth) for r in range(world)]
    shared = torch.cuda.Stream("cuda:0")  # one stream orders all subdomains and the copies below
    solvers = [fdist.HipSubdomainSolver(s, g.pos[s.vid], s.edges, g.alpha[s.eid], g.beta[s.eid],
                                        g.z[s.vid], g.wgt[s.vid], None, device=0, stream=shared)
               for s in subs]
    # request lists, as PartitionedSolver builds them through all_gather_object
    send = {(o, r): ([], []) for o in range(world) for r in range(world)}
    for r, s in enumerate(subs):
        g2l = None
        for o, v in s.recv_v.items():
            send[(o, r)] = (s.vid[v].tolist(), send[(o, r)][1])
        for o, e in s.recv_e.items():
            send[(o, r)] = (send[(o, r)][0], s.eid[e].tolist())
    offs = []
    for r, s in enumerate(subs):
        sv, se, rv, re_ = [], [], [], []
        for o in range(world):
            if o == r:
                continue
            # global ids -> this rank's local numbering by binary search (both lists ascending)
            sv += np.searchsorted(s.vid[:s.n_own], np.asarray(send[(r, o)][0], np.int64)).tolist()
            se += np.searchsorted(s.eid, np.asarray(send[(r, o)][1], np.int64)).tolist()
            rv += s.recv_v.get(o, np.zeros(0, np.int32)).tolist()
            re_ += s.recv_e.get(o, np.zeros(0, np.int32)).tolist()
        i32 = lambda a: np.asarray(a, np.int32)  # noqa: E731
        solvers[r].halo_register(i32(sv), i32(se), i32(rv), i32(re_))
        offs.append({o: (len(send[(r, o)][0]), len(send[(r, o)][1])) for o in range(world) if o != r})

    def slices(cnt_by_peer, peers):
        nv = sum(c[0] for c in cnt_by_peer.values())
        out, ov, oe = {}, 0, fdist.VREC * nv
        for o in peers:
            a, b = cnt_by_peer[o]
            out[o] = ((ov, ov + fdist.VREC * a), (oe, oe + fdist.EREC * b))
            ov += fdist.VREC * a
            oe += fdist.EREC * b
        return out

    p = default_params()
    done = 0
    while done < iters:
        n = min(depth, iters - done)
        for sv in solvers:
            sv.step(p, n)
        done += n
        if done >= iters:
            break
        packed = [sv.halo_pack() for sv in solvers]
        for r in range(world):
          with torch.cuda.stream(shared):
              peers = [o for o in range(world) if o != r]
              recv_cnt = {o: offs[o][r] for o in peers}          # what o sends to r
              rbuf = torch.empty(fdist.VREC * sum(c[0] for c in recv_cnt.values()) + fdist.EREC * sum(c[1] for c in recv_cnt.values()),
                                 dtype=torch.float32, device="cuda:0")
              rs = slices(recv_cnt, peers)
              for o in peers:
                  ss = slices(offs[o], [q for q in range(world) if q != o])[r]
                  for (a, b), (c, d) in zip(ss, rs[o]):
                      rbuf[c:d] = packed[o][a:b]
              solvers[r].halo_unpack(rbuf)
    return subs, solvers
