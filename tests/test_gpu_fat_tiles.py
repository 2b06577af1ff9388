"""-m gpu: FAT resident tiles (r05) -- a graph beyond 256 x 196 vertices still gets one tile per CU and ONE launch per solve:
the halo gets shallower as the tiles grow, the outermost ring has no lanes of its own once a tile holds more local vertices
than threads, and the incidence slots shrink to 12 bytes (two naturally aligned arrays) where 16 do not fit 160 KiB.  Every
variant against the oracle, bit for bit; a resident launch that gives up is repeated by launches on the same plan."""
import os
import subprocess
import sys

import numpy as np
import pytest

from flame_ros_amd.regularizer import GraphRegularizer, default_params
from tests.util import assert_bit_equal, graphgen, hooks_env, make_oracle, oracle_params, with_hooks

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (vertices, expected halo depth, 12-byte slots, may a tile hold more local vertices than threads)
FAT = [(60000, 4, 0), (100000, 3, 0), (135000, 2, 0), (160000, 1, 0), (200000, 2, 1)]


@pytest.mark.parametrize("V,depth,slot12", FAT)
def test_fat_resident_tiles_match_oracle(gpu, V, depth, slot12):
    g = graphgen.synthetic(V, seed=V)
    iters = 90
    o = make_oracle(g)
    o.solve(oracle_params(), iters)
    with GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0) as r:
        assert r.info("num_tiles") == 256 and r.info("tile_depth") == depth and r.info("tile_slot12") == slot12
        assert r.info("tile_lds_bytes") <= 160 * 1024
        r.step(default_params(), 40)   # (a plan's first solve: poll lists in local order, sorted local edges)
        assert r.info("persist_used") == 1 and r.last_solve_ms()[1] == 1
        r.step(default_params(), iters - 40)  # (second solve: address-sorted poll lists, lane order)
        assert r.info("persist_used") == 1 and r.last_solve_ms()[1] == 1
        x, w1, w2, q = r.download()
        assert r.info("persist_recovered") == 0
    assert_bit_equal(x, o.x, "x"); assert_bit_equal(w1, o.w1, "w1"); assert_bit_equal(w2, o.w2, "w2"); assert_bit_equal(q, o.q, "q")


def test_fat_tiles_with_more_local_vertices_than_threads(gpu):
    """200 k vertices: the largest tiles hold > 1024 local vertices on 1024 threads -- only the updated vertices and the poll
    list's slots need a lane."""
    g = graphgen.synthetic(200000, seed=200000)
    with GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0) as r:
        import ctypes as C
        from flame_ros_amd import lib
        raw = r.plan_array("tiles", np.dtype((np.void, C.sizeof(lib.TileDesc))))
        tiles = (lib.TileDesc * len(raw)).from_buffer_copy(raw.tobytes())
        nt = r.info("tile_threads") * r.info("tile_vpt")
        assert max(t.n_ext for t in tiles) > nt
        assert max(t.n_upd for t in tiles) <= nt and max(t.n_ext - t.n_own for t in tiles) <= nt
        assert max(t.e_loc for t in tiles) <= r.info("tile_threads") * r.info("tile_ept")


def test_without_resident_tiles_the_two_round_partition_stays(gpu):
    g = graphgen.synthetic(100000, seed=100000)
    o = make_oracle(g)
    o.solve(oracle_params(), 30)
    with GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, persist=0) as r:
        assert r.info("num_tiles") > 256 and r.info("tile_slot12") == 0
        r.step(default_params(), 30)
        assert r.info("persist_used") == 0
        assert_bit_equal(r.download()[0], o.x, "x")


@pytest.mark.parametrize("V,slot12", [(200000, 1), (100000, 0)])
def test_a_fat_resident_solve_that_gives_up_is_repeated_by_launches(gpu, V, slot12):
    """Test hook persist_fail (hooks library: every resident launch counts as failed): the solve is repeated by ordinary launches of the SAME
    fat plan -- 12-byte slots and lane-less outer ring through k_tile -- with the oracle's bits."""
    code = r'''
import numpy as np, sys
sys.path.insert(0, %r)
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params
from oracle import COracle
from oracle.cbind import default_params as oparams
for V, s12 in ((%d, %d),):  # (one graph per process: a give-up starts the process-wide back-off)
    g = graphgen.synthetic(V, seed=V)
    o = COracle(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt)
    o.solve(oparams(), 25)
    r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0)
    assert r.info("tile_slot12") == s12 and r.info("num_tiles") == 256
    r.step(default_params(), 25, sync=False)
    assert r.info("persist_used") == 1
    x, w1, w2, q = r.download()
    assert r.info("persist_recovered") == 1, r.info("persist_recovered")
    for a, b, n in ((x, o.x, "x"), (w1, o.w1, "w1"), (w2, o.w2, "w2"), (q, o.q, "q")):
        assert np.array_equal(a.view(np.uint32), np.asarray(b, np.float32).view(np.uint32)), (V, n)
    r.close()
print("ok")
''' % (ROOT, V, slot12)
    out = subprocess.run([sys.executable, "-c", with_hooks(code, persist_fail=1)], env=hooks_env(), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
