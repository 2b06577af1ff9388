"""-m gpu: option "persist" -- graphs of <= 32 tiles solved by ONE launch of resident tiles on one XCD
(kernels.hip k_tile_persist: a counter barrier and a re-read of the halo state through L2 instead of a
kernel boundary per `depth` iterations).  Same bits as the launches per round and as the oracle."""
import numpy as np
import pytest

from flame_ros_amd.regularizer import GraphRegularizer, default_params
from tests.util import assert_bit_equal, graphgen, make_oracle, oracle_params

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,own,depth", [("tum", 40, 5), ("tum", 64, 4), ("v2000", 70, 5), ("v800", 30, 8), ("tum", 40, 2)])
def test_persistent_tiles_match_oracle(gpu, name, own, depth):
    g, _ = graphgen.named(name)
    p = default_params()
    r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, tile_own=own, tile_depth=depth,
                         persist=1)
    ref = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, tile_own=own, tile_depth=depth)
    assert 2 <= r.info("num_tiles") <= 32 and r.info("tile_depth") == depth
    o = make_oracle(g)
    for iters in (depth + 1, 200, 23, depth, 1, 77):  # ragged last rounds; <= depth: the ordinary launch
        o.solve(oracle_params(), iters)
        r.step(p, iters); ref.step(p, iters)
        assert r.info("persist_used") == (1 if iters > depth else 0), iters
        assert ref.info("persist_used") == 0
        x, w1, w2, q = r.download()
        xr, w1r, w2r, qr = ref.download()
        assert_bit_equal(x, o.x, "%s %d x" % (name, iters)); assert_bit_equal(q, o.q, "%s %d q" % (name, iters))
        assert_bit_equal(w1, o.w1, "w1"); assert_bit_equal(w2, o.w2, "w2")
        assert_bit_equal(x, xr, "vs launches x"); assert_bit_equal(q, qr, "vs launches q")
    xb = r.download_bar()
    xbr = ref.download_bar()
    for a, b in zip(xb, xbr):
        assert_bit_equal(a, b, "x_bar")
    r.close(); ref.close()


def test_persist_not_taken_on_larger_graphs(gpu):
    g, it = graphgen.named("5k")
    r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, persist=1)
    assert r.info("num_tiles") > 32
    r.step(default_params(), 50)
    assert r.info("persist_used") == 0
    o = make_oracle(g); o.solve(oracle_params(), 50)
    assert_bit_equal(r.download()[0], o.x, "x")
    r.close()
