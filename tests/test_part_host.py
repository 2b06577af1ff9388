"""CPU: the library's own partition mode (csrc/part.cpp, include/flame_hip.h flame_hip_part_*) as a host-only plan.

The C++ subdomain construction must equal flame_ros_amd/dist.py's (the scheme the world-2 / world-3 gloo tests prove
bit-exact) array for array, the message lists of every pair of parts must agree between the sending and the receiving
rank, and librccl.so must load with every entry point the exchange uses (the link check of VERDICT r03 item 6)."""
import numpy as np
import pytest

from flame_ros_amd import dist as fdist
from flame_ros_amd import lib, partition
from tests.util import graphgen


def test_rccl_loads_with_every_entry_point():
    assert partition.rccl_available(), "librccl.so.1 did not load or lacks ncclSend / ncclRecv / ncclGroup* / ncclAllReduce"


def test_comm_needs_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(lib.FlameHipError) as e:
        partition.Communicator(0, 0, 1, b"\0" * 128)
    assert e.value.code == lib.ERR_NODEVICE


@pytest.mark.parametrize("world,k,depth,V", [(2, 1, 4, 1500), (3, 1, 2, 1500), (1, 3, 4, 1800), (2, 3, 2, 1800), (8, 1, 16, 20000)])
def test_host_plan_equals_the_python_harness(world, k, depth, V):
    g = graphgen.synthetic(V, seed=11)
    nparts = world * k
    part_py = fdist.rcb_parts(g.pos, nparts)
    plans = [partition.Partition(None, g.pos, g.edges, None, None, None, None, parts_per_rank=k, halo_depth=depth, plan_rank=r,
                                 plan_world=world) for r in range(world)]
    sends, recvs = {}, {}
    for r, P in enumerate(plans):
        assert np.array_equal(P.array("part"), part_py)
        assert P.info("num_parts") == nparts
        for i in range(k):
            me = r * k + i
            sub = fdist.build_subdomain(g.pos, g.edges, part_py, me, depth)
            assert P.info("part_id", i) == me and P.info("n_own", i) == sub.n_own
            vid, eid = P.array("vid", i), P.array("eid", i)
            assert np.array_equal(vid, sub.vid) and np.array_equal(eid, sub.eid)
            assert np.array_equal(P.array("edges", i).reshape(-1, 2), sub.edges)
            assert np.array_equal(P.array("e_owned", i).astype(bool), sub.e_owned)
            peers = P.array("peers", i)
            assert list(peers) == sorted(peers) and me not in peers
            sc, rc = P.array("send_cnt", i).reshape(-1, 2), P.array("recv_cnt", i).reshape(-1, 2)
            sv, se, rv, re_ = P.array("send_v", i), P.array("send_e", i), P.array("recv_v", i), P.array("recv_e", i)
            assert sv.max(initial=-1) < sub.n_own  # only own vertices are sent
            ov = oe = iv = ie = 0
            for j, p in enumerate(peers):
                sends[(me, int(p))] = (vid[sv[ov:ov + sc[j, 0]]], eid[se[oe:oe + sc[j, 1]]])
                recvs[(int(p), me)] = (vid[rv[iv:iv + rc[j, 0]]], eid[re_[ie:ie + rc[j, 1]]])
                # the receive lists are the python harness's, peer by peer
                assert np.array_equal(rv[iv:iv + rc[j, 0]], sub.recv_v.get(int(p), np.zeros(0, np.int32)))
                assert np.array_equal(re_[ie:ie + rc[j, 1]], sub.recv_e.get(int(p), np.zeros(0, np.int32)))
                ov += sc[j, 0]; oe += sc[j, 1]; iv += rc[j, 0]; ie += rc[j, 1]
            assert ov == len(sv) and oe == len(se) and iv == len(rv) and ie == len(re_)
            assert set(int(p) for p in peers) >= set(sub.recv_v) | set(sub.recv_e)
    # what part a sends to part b is exactly what b expects from a, in the same order (messages match by order)
    assert set(sends) == set(recvs)
    for key in sends:
        assert np.array_equal(sends[key][0], recvs[key][0]) and np.array_equal(sends[key][1], recvs[key][1]), key
    for P in plans:
        P.close()


def test_bad_arguments():
    g = graphgen.synthetic(300, seed=2)
    for kw in (dict(parts_per_rank=0), dict(halo_depth=0), dict(halo_depth=65), dict(plan_rank=2, plan_world=2)):
        with pytest.raises(lib.FlameHipError) as e:
            partition.Partition(None, g.pos, g.edges, None, None, None, None, **kw)
        assert e.value.code == lib.ERR_ARG
    bad = g.edges.copy(); bad[0, 0] = 300
    with pytest.raises(lib.FlameHipError):
        partition.Partition(None, g.pos, bad, None, None, None, None)
