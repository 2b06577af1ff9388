"""-m gpu: the BASELINE.json configurations that round 1 left untested on the GPU box, the committed
golden fixtures read directly, and the frame-stream use of ONE handle.

  config 3  EuRoC-shaped graph at its stated size (~10 k vertices), 200 iterations
  config 5  synthetic 200 k-vertex / 600 k-edge graph, 500 iterations: single GPU, and cut 8-way
            into subdomains (halo depth 16) whose exchange runs through the real pack/unpack
            kernels on one GPU (what RCCL moves between 8 GPUs)
All comparisons are bit-exact against the CPU oracle (oracle/nltgv2_oracle.c).
"""
import os

import numpy as np
import pytest

from flame_ros_amd.regularizer import GraphRegularizer, default_params, default_tri_params
from oracle.cbind import depthmaps as oracle_depthmaps, mesh as oracle_mesh, triangles as oracle_triangles, TriParams as OTri
from tests.halo_driver import run_subdomains_one_gpu
from tests.util import assert_bit_equal, graphgen, make_oracle, oracle_params

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def g200k():
    """Config 5 graph + the oracle's state after the full 500 iterations (~4 s of CPU)."""
    g, iters = graphgen.named("200k")
    assert g.V == 200000 and iters == 500
    o = make_oracle(g)
    o.solve(oracle_params(), iters)
    return g, iters, o


def test_config5_200k_single_gpu(gpu, g200k):
    g, iters, o = g200k
    with GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris) as r:
        assert r.info("path") == 2  # the tile path is the product path at this size
        r.step(default_params(), iters)
        x, w1, w2, q = r.download()
        sg, dg = r.costs(default_params())
    assert_bit_equal(x, o.x, "200k x")
    assert_bit_equal(w1, o.w1, "200k w1")
    assert_bit_equal(w2, o.w2, "200k w2")
    assert_bit_equal(q, o.q, "200k q")
    so, do = o.costs(oracle_params())
    assert abs(sg - so) <= 1e-9 * so and abs(dg - do) <= 1e-9 * do
    assert float(np.sqrt(np.mean((x.astype(np.float64) - o.x) ** 2))) <= 1e-4  # north_star tolerance


def test_config5_200k_eight_subdomains(gpu, g200k):
    """BASELINE config 5: 8-way partition, halo depth 16; all eight subdomains on this one GPU."""
    g, iters, o = g200k
    subs, solvers = run_subdomains_one_gpu(g, 8, 16, iters)
    owned = np.zeros(g.V, bool)
    for r, s in enumerate(subs):
        x, w1, w2, q = solvers[r].download()
        own = slice(0, s.n_own)
        assert_bit_equal(x[own], o.x[s.vid[own]], "subdomain %d x" % r)
        assert_bit_equal(w1[own], o.w1[s.vid[own]], "subdomain %d w1" % r)
        assert_bit_equal(w2[own], o.w2[s.vid[own]], "subdomain %d w2" % r)
        oe = np.flatnonzero(s.e_owned)
        assert_bit_equal(q[oe], o.q[s.eid[oe]], "subdomain %d q" % r)
        owned[s.vid[own]] = True
    assert owned.all()
    for sv in solvers:
        sv.reg.close()


def test_config4_50k_two_subdomains_full_size(gpu):
    """BASELINE config 4 at FULL size: the 50 000-vertex graph cut 2-way (RCB; METIS is absent from the
    image), halo depth 16, 500 PD iterations, both subdomains through the real kernels on this one GPU
    (what RCCL moves between two MI355X travels through device buffers here): own state bit-exact
    against the oracle, and the owned-cost sums add up to the whole graph's costs (the all-reduce of
    2 doubles of partition mode)."""
    g, iters = graphgen.named("50k")
    assert g.V == 50000 and iters == 500
    o = make_oracle(g)
    o.solve(oracle_params(), iters)
    subs, solvers = run_subdomains_one_gpu(g, 2, 16, iters)
    sm = da = 0.0
    for r, s in enumerate(subs):
        x, w1, w2, q = solvers[r].download()
        own = slice(0, s.n_own)
        assert 24000 < s.n_own < 26000 and len(s.vid) - s.n_own > 1000  # a real cut with a deep halo
        assert_bit_equal(x[own], o.x[s.vid[own]], "subdomain %d x" % r)
        assert_bit_equal(w1[own], o.w1[s.vid[own]], "subdomain %d w1" % r)
        oe = np.flatnonzero(s.e_owned)
        assert_bit_equal(q[oe], o.q[s.eid[oe]], "subdomain %d q" % r)
        # 500 = 31 x 16 + 4: the last chunk left 12 halo rings exact, so owned edges see exact targets
        vmask = np.zeros(len(s.vid), np.uint8)
        vmask[own] = 1
        a, b = solvers[r].costs_owned(default_params(), vmask, s.e_owned.astype(np.uint8))
        sm += a
        da += b
    so, do = o.costs(oracle_params())
    assert abs(sm - so) <= 1e-9 * so and abs(da - do) <= 1e-9 * do, (sm, so, da, do)
    for sv in solvers:
        sv.reg.close()


def test_config3_euroc_10k(gpu):
    g, iters = graphgen.named("euroc")
    assert g.V == 10000
    o = make_oracle(g)
    o.solve(oracle_params(), iters)
    for opts in ({}, dict(path=1)):
        with GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, **opts) as r:
            r.step(default_params(), iters)
            x, w1, w2, q = r.download()
        assert_bit_equal(x, o.x, "euroc x %s" % opts)
        assert_bit_equal(q, o.q, "euroc q %s" % opts)


def test_golden_fixtures_direct(gpu):
    """The HIP path against the committed fixtures themselves (tests/golden/*.npz), no live oracle
    in the loop: x after 1 / 10 / 200 iterations of the 5 k graph, the 12-vertex graph, K4."""
    d = np.load(os.path.join(GOLDEN, "g5k.npz"))
    for opts in ({}, dict(path=1), dict(tile_own=64, tile_depth=3)):
        with GraphRegularizer(d["pos"], d["edges"], d["alpha"], d["beta"], d["z"], d["wgt"], **opts) as r:
            done = 0
            for n in (1, 10, 200):
                r.step(default_params(), n - done)
                done = n
                assert_bit_equal(r.download(with_q=False)[0], d["x_after_%d" % n], "g5k x after %d %s" % (n, opts))
            x, w1, w2, q = r.download()
            assert_bit_equal(w1, d["w1_after_200"], "g5k w1")
            assert_bit_equal(w2, d["w2_after_200"], "g5k w2")
            assert_bit_equal(q, d["q_after_200"], "g5k q")
            s, dc = r.costs(default_params())
            assert np.allclose([s, dc], d["costs_after_200"], rtol=1e-9, atol=0)
    d = np.load(os.path.join(GOLDEN, "g12.npz"))
    with GraphRegularizer(d["pos"], d["edges"], d["alpha"], d["beta"], d["z"], d["wgt"]) as r:
        r.step(default_params(), 1)
        assert_bit_equal(r.download(with_q=False)[0], d["x_after_1"], "g12 x after 1")
        r.step(default_params(), 4)
        x, w1, w2, q = r.download()
        assert_bit_equal(x, d["x_after_5"], "g12 x after 5")
        assert_bit_equal(w1, d["w1_after_5"], "g12 w1")
        assert_bit_equal(q, d["q_after_5"], "g12 q")


def test_frame_stream_on_one_handle(gpu):
    """FLaME re-triangulates every frame (reference src/flame_offline_tum.cc:578): ONE handle is
    resized and re-uploaded for 20 frames of varying size -- including a frame without triangles,
    a 3-vertex frame and growth past the first capacity -- and every frame is bit-exact."""
    sizes = [3000, 3100, 2950, 1200, 400, 3, 5200, 5150, 9000, 700, 3050, 3051, 12000, 64, 2999,
             3000, 1500, 1501, 8000, 3000]
    r = None
    K = np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]])
    Kinv = np.linalg.inv(K).astype(np.float32)
    tp = default_tri_params(640, 480)
    otp = OTri(*[getattr(tp, f[0]) for f in tp._fields_])
    for k, V in enumerate(sizes):
        g = graphgen.synthetic(V, seed=100 + k)
        tris = None if k == 3 else g.tris
        iters = 37 + k
        if r is None:
            r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=tris)
        else:
            r.reupload(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=tris)
        assert (r.info("V"), r.info("E"), r.info("T")) == (g.V, g.E, 0 if tris is None else g.T)
        o = make_oracle(g)
        o.solve(oracle_params(), iters)
        r.step(default_params(), iters)
        x, w1, w2, q = r.download()
        assert_bit_equal(x, o.x, "frame %d (V=%d) x" % (k, V))
        assert_bit_equal(q, o.q, "frame %d (V=%d) q" % (k, V))
        if tris is not None:
            tn_o, tv_o, vn_o = oracle_triangles(otp, Kinv, g.pos, o.x, g.tris)
            tn, tv, vn = r.triangles(Kinv, tp)
            assert np.array_equal(tv, tv_o)
            assert_bit_equal(vn, vn_o, "frame %d normals" % k)
    r.close()


def test_graph_without_triangles(gpu):
    """T == 0: the triangle stage, the mesh and the dense maps run on a graph that
    has vertices but no triangles: degenerate normals (0,0,-1), no faces, nothing covered."""
    g = graphgen.synthetic(800, seed=7)
    o = make_oracle(g)
    o.solve(oracle_params(), 20)
    K = np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]])
    Kinv = np.linalg.inv(K).astype(np.float32)
    tp = default_tri_params(640, 480)
    otp = OTri(*[getattr(tp, f[0]) for f in tp._fields_])
    none = np.zeros((0, 3), np.int32)
    with GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=None) as r:
        r.step(default_params(), 20)
        assert_bit_equal(r.download(with_q=False)[0], o.x, "x")
        tn, tv, vn = r.triangles(Kinv, tp)
        _, _, vn_o = oracle_triangles(otp, Kinv, g.pos, o.x, none)
        assert tn.shape == (0, 3) and tv.shape == (0,)
        assert_bit_equal(vn, vn_o, "vertex normals without triangles")
        assert np.array_equal(vn, np.tile(np.float32([0, 0, -1]), (g.V, 1)))
        pts, faces = r.mesh(Kinv, tp)
        pts_o, faces_o = oracle_mesh(Kinv, g.pos, o.x, vn_o, none, np.zeros(0, np.uint8), 640, 480)
        assert faces.shape == (0, 3) and len(faces_o) == 0
        assert np.array_equal(pts.view(np.uint32), pts_o.view(np.uint32))
        idm, dm, cl = r.depthmaps(Kinv, tp)
        idm_o, _, _ = oracle_depthmaps(640, 480, g.pos, o.x, none, np.zeros(0, np.uint8), True, Kinv, 0.1, 100.0)
        assert np.isnan(idm).all() and np.isnan(dm).all() and np.isnan(cl).all()
        assert np.isnan(idm_o).all()
