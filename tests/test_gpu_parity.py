"""-m gpu: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

The oracle restates upstream's sequential step() (parity vs upstream itself is
UNPINNED, see oracle/nltgv2_oracle.h); the arithmetic contract makes every solver path and every
partitioning produce the oracle's exact float32 bits, so the tolerance here is zero.  The
north_star tolerance (1e-4 RMS idepth) is asserted as well, trivially.
"""
import numpy as np
import pytest

from flame_ros_amd.regularizer import GraphRegularizer, default_params, default_tri_params
from oracle.cbind import depthmaps as oracle_depthmaps, mesh as oracle_mesh, triangles as oracle_triangles, TriParams as OTri
from tests.util import assert_bit_equal, graphgen, make_oracle, oracle_params, random_state

pytestmark = pytest.mark.gpu

PATHS = {
    "global": dict(path=1),
    "tile_auto": dict(path=2),
    "tile_small": dict(path=2, tile_own=64, tile_depth=2),
    "tile_deep": dict(path=2, tile_own=256, tile_depth=6),
    "tile_nograph": dict(path=2, use_graph=0),
    "tile_hostplan": dict(path=2, plan_device=0),
}


def run_both(g, opts, iters, state_seed=None, chunks=None):
    o = make_oracle(g)
    r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, **opts)
    if state_seed is not None:
        st = random_state(g, state_seed)
        o.set_state(**st)
        r.set_state(**st)
    po, pg = oracle_params(), default_params()
    for n in (chunks or [iters]):
        o.solve(po, n)
        r.step(pg, n)
    return o, r


def compare_state(o, r, what):
    x, w1, w2, q = r.download()
    xb, w1b, w2b = r.download_bar()
    assert_bit_equal(x, o.x, what + " x")
    assert_bit_equal(w1, o.w1, what + " w1")
    assert_bit_equal(w2, o.w2, what + " w2")
    assert_bit_equal(q, o.q, what + " q")
    assert_bit_equal(xb, o.xb, what + " xb")
    assert_bit_equal(w1b, o.w1b, what + " w1b")
    assert_bit_equal(w2b, o.w2b, what + " w2b")
    rms = float(np.sqrt(np.mean((x.astype(np.float64) - o.x) ** 2)))
    assert rms <= 1e-4  # north_star tolerance


@pytest.mark.parametrize("path", list(PATHS))
@pytest.mark.parametrize("V,iters", [(1200, 50), (5000, 200)])
def test_solve_matches_oracle(gpu, path, V, iters):
    g = graphgen.synthetic(V, seed=1)
    o, r = run_both(g, PATHS[path], iters)
    compare_state(o, r, "%s V=%d" % (path, V))
    if path.startswith("tile"):
        assert r.info("path") == 2


@pytest.mark.parametrize("path", ["global", "tile_auto", "tile_small"])
def test_random_state_and_ragged_iteration_counts(gpu, path):
    """Non-trivial initial state; iteration counts that are not multiples of the tile depth."""
    g = graphgen.synthetic(3000, seed=2)
    o, r = run_both(g, PATHS[path], None, state_seed=3, chunks=[1, 2, 3, 5, 7, 0, 13])
    compare_state(o, r, path)


def test_dataset_shaped_graphs(gpu):
    """Config 1 / 3 stand-ins: TUM 640x480 @ win 16 (one LDS tile), EuRoC 752x480 @ win 8."""
    for (w, h, win, iters) in ((640, 480, 16, 200), (752, 480, 8, 200)):
        g = graphgen.dataset_shaped(w, h, win)
        o, r = run_both(g, {}, iters)
        compare_state(o, r, "win%d" % win)
    g = graphgen.dataset_shaped(640, 480, 16)  # the same frame forced into one isolated LDS tile
    o, r = run_both(g, dict(tile_own=g.V), 200)
    assert r.info("num_tiles") == 1
    compare_state(o, r, "single tile")


def test_costs_match_oracle(gpu):
    g = graphgen.synthetic(5000, seed=4)
    o, r = run_both(g, {}, 100)
    so, do = o.costs(oracle_params())
    sg, dg = r.costs(default_params())
    assert abs(sg - so) <= 1e-9 * abs(so) and abs(dg - do) <= 1e-9 * abs(do)
    assert r.smoothnessCost(default_params()) == sg and r.dataCost(default_params()) == dg


def test_triangle_stage_matches_oracle(gpu):
    g = graphgen.synthetic(5000, seed=5)
    o, r = run_both(g, {}, 50)
    K = np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]])
    Kinv = np.linalg.inv(K).astype(np.float32)
    tp = default_tri_params(640, 480)
    otp = OTri(*[getattr(tp, f[0]) for f in tp._fields_])
    tn_o, tv_o, vn_o = oracle_triangles(otp, Kinv, g.pos, o.x, g.tris)
    tn, tv, vn = r.triangles(Kinv, tp)
    assert_bit_equal(tn, tn_o, "tri normals")
    assert np.array_equal(tv, tv_o)
    assert_bit_equal(vn, vn_o, "vertex normals")
    assert 0 < tv.sum() < len(tv)  # the filters reject some but not all triangles


def test_full_size_50k_properties(gpu):
    """BASELINE config 4 size: oracle comparison on the full 500 iterations + size-independent
    properties (determinism across paths, dual feasibility, energy decrease)."""
    g, iters = graphgen.named("50k")
    o, r = run_both(g, {}, iters)
    compare_state(o, r, "50k")
    _, r2 = run_both(g, PATHS["global"], 0)
    r2.step(default_params(), iters)
    assert_bit_equal(r2.download()[0], r.download()[0], "tile vs global path")
    q = r.download()[3]
    assert np.all(np.abs(q) <= 1.0)
    s1, d1 = r.costs(default_params())
    r0 = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt)
    s0, d0 = r0.costs(default_params())
    assert s1 + d1 < s0 + d0


def test_edge_cases(gpu):
    p = default_params()
    # empty graph, single vertex, single edge, two components
    r = GraphRegularizer(np.zeros((0, 2)), np.zeros((0, 2), np.int32), [], [], [], [])
    r.step(p, 5)
    assert r.download()[0].shape == (0,)
    r = GraphRegularizer([[1.0, 2.0]], np.zeros((0, 2), np.int32), [], [], [0.5], [1.0])
    r.step(p, 5)
    assert r.download()[0][0] == np.float32(0.5)
    g = graphgen.synthetic(40, seed=7)
    for opts in (dict(path=1), dict(path=2), dict(path=2, tile_own=8, tile_depth=3)):
        o, r = run_both(g, opts, 33, state_seed=8)
        compare_state(o, r, "tiny %s" % opts)
    # mostly isolated vertices (degree 0) around a small connected part, several tiles
    g = graphgen.synthetic(120, seed=9)
    rng = np.random.default_rng(10)
    extra = 1500
    g.pos = np.concatenate([g.pos, rng.uniform(0, 640, (extra, 2)).astype(np.float32)])
    g.z = np.concatenate([g.z, rng.uniform(0.2, 1.0, extra).astype(np.float32)])
    g.wgt = np.ones(len(g.z), np.float32)
    g.tris = None
    for opts in (dict(path=1), dict(path=2, tile_own=50, tile_depth=3), dict(path=2, tile_own=10 ** 6)):
        o, r = run_both(g, opts, 21, state_seed=11)
        compare_state(o, r, "isolated %s" % opts)


def test_batch_of_frames(gpu):
    """Frames axis: a batch of independent feature graphs in one handle (one LDS tile = one
    workgroup per frame, every iteration in a single launch); each frame equals its own oracle."""
    gs = [graphgen.dataset_shaped(640, 480, 16, seed=s) for s in range(6)] + \
         [graphgen.synthetic(300, seed=9), graphgen.dataset_shaped(320, 240, 8, seed=4)]
    r = GraphRegularizer.from_batch(gs)
    assert r.info("num_tiles") == len(gs) and r.info("tile_depth") == 0
    r.step(default_params(), 150)
    ms, launches = r.last_solve_ms()
    assert launches == 1
    x, w1, w2, q = r.download()
    eoff = np.cumsum([0] + [g.E for g in gs])
    for b, g in enumerate(gs):
        o = make_oracle(g)
        o.solve(oracle_params(), 150)
        sl = slice(r.voff[b], r.voff[b + 1])
        assert_bit_equal(x[sl], o.x, "frame %d x" % b)
        assert_bit_equal(w1[sl], o.w1, "frame %d w1" % b)
        assert_bit_equal(q[eoff[b]:eoff[b + 1]], o.q, "frame %d q" % b)
    # the batch solved again (the lane order of the plan is re-assigned on the device before the
    # second solve, option lane_order = 1) and a third time from the captured launch
    emap0 = r.plan_array("t_emap", np.int32)
    for _ in range(2):
        r.step(default_params(), 50)
    assert not np.array_equal(emap0, r.plan_array("t_emap", np.int32))
    x, w1, w2, q = r.download()
    for b, g in enumerate(gs):
        o = make_oracle(g)
        o.solve(oracle_params(), 250)
        assert_bit_equal(x[r.voff[b]:r.voff[b + 1]], o.x, "frame %d x, resolved" % b)
        assert_bit_equal(q[eoff[b]:eoff[b + 1]], o.q, "frame %d q, resolved" % b)


def test_mesh_points_and_faces(gpu):
    """Row f1: PointNormalUV vertices + reversed-winding faces, incl. invalid (NaN) vertices."""
    g = graphgen.synthetic(4000, seed=6)
    o, r = run_both(g, {}, 30)
    bad = np.arange(0, g.V, 97)
    x = o.x.copy()
    x[bad[::2]] = np.nan
    x[bad[1::2]] = -0.5
    o.set_state(x=x)
    r.set_state(x=x)
    K = np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]])
    Kinv = np.linalg.inv(K).astype(np.float32)
    tp = default_tri_params(640, 480)
    otp = OTri(*[getattr(tp, f[0]) for f in tp._fields_])
    _, tv_o, vn_o = oracle_triangles(otp, Kinv, g.pos, o.x, g.tris)
    pts_o, faces_o = oracle_mesh(Kinv, g.pos, o.x, vn_o, g.tris, tv_o, 640, 480)
    pts, faces = r.mesh(Kinv, tp)
    assert np.array_equal(faces, faces_o) and 0 < len(faces) < g.T
    assert np.array_equal(np.isnan(pts), np.isnan(pts_o)) and np.isnan(pts[bad, :3]).all()
    ok = ~np.isnan(pts_o)
    assert_bit_equal(pts[ok], pts_o[ok], "mesh points")
    assert np.all(pts[bad, 3:] == 0)


@pytest.mark.parametrize("filtered", [True, False])
def test_dense_maps(gpu, filtered):
    """Row f2: idepthmap rasterisation, idepth -> depth inversion, point cloud."""
    g = graphgen.dataset_shaped(640, 480, 16, seed=7)
    o, r = run_both(g, {}, 40)
    x = o.x.copy()
    x[::53] = np.nan
    x[7::61] = -1.0
    o.set_state(x=x)
    r.set_state(x=x)
    K = np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]])
    Kinv = np.linalg.inv(K).astype(np.float32)
    tp = default_tri_params(640, 480)
    otp = OTri(*[getattr(tp, f[0]) for f in tp._fields_])
    _, tv_o, _ = oracle_triangles(otp, Kinv, g.pos, o.x, g.tris)
    idm_o, dm_o, cl_o = oracle_depthmaps(640, 480, g.pos, o.x, g.tris, tv_o, filtered, Kinv, 0.1, 100.0)
    idm, dm, cl = r.depthmaps(Kinv, tp, filtered=filtered, min_depth=0.1, max_depth=100.0)
    for got, want, name in ((idm, idm_o, "idepthmap"), (dm, dm_o, "depthmap"), (cl, cl_o, "cloud")):
        assert np.array_equal(np.isnan(got), np.isnan(want)), name
        ok = ~np.isnan(want)
        assert_bit_equal(got[ok], want[ok], name)
    cover = 1.0 - np.isnan(idm).mean()
    assert 0.5 < cover <= 1.0 and (np.isnan(dm).sum() >= np.isnan(idm).sum())
    if filtered:
        assert np.isnan(idm).sum() > np.isnan(r.depthmaps(Kinv, tp, filtered=False, cloud=False)[0]).sum()


def test_irregular_graphs(gpu):
    """Degree distributions no Delaunay graph has: a 3000-leaf star (one incidence list too long
    for any LDS tile -> automatic global path), a path, parallel edges and both orientations."""
    rng = np.random.default_rng(3)

    class G:
        pass

    def mk(pos, edges):
        g = G()
        g.pos = np.asarray(pos, np.float32)
        g.edges = np.asarray(edges, np.int32)
        g.V, g.E = len(g.pos), len(g.edges)
        d = g.pos[g.edges[:, 0]] - g.pos[g.edges[:, 1]]
        g.alpha = (1.0 / np.maximum(np.sqrt((d ** 2).sum(1)), 1.0)).astype(np.float32)
        g.beta = g.alpha.copy()
        g.z = (0.5 + 0.3 * rng.random(g.V)).astype(np.float32)
        g.wgt = np.ones(g.V, np.float32)
        g.tris = None
        return g

    n = 3000
    star = mk(np.c_[rng.uniform(0, 640, n + 1), rng.uniform(0, 480, n + 1)],
              [(0, k) if k % 2 else (k, 0) for k in range(1, n + 1)])
    path = mk(np.c_[np.arange(4000) * 3.0, np.zeros(4000)], [(k, k + 1) for k in range(3999)])
    multi = mk(rng.uniform(0, 100, (50, 2)),
               [(a, b) for a in range(50) for b in range(a + 1, 50) if (a + b) % 7 == 0] * 2)
    for name, g, want_path in (("star", star, 1), ("path", path, 2), ("multi", multi, 2)):
        o, r = run_both(g, {}, 60, state_seed=12)
        assert r.info("path") == want_path, name
        compare_state(o, r, name)


def test_distinct_handles_are_thread_safe_and_deterministic(gpu):
    """include/flame_hip.h: one handle is not thread-safe, distinct handles are.  Four host
    threads drive four handles concurrently (ctypes drops the GIL); every result equals the
    oracle, and a repeated solve reproduces the same bits."""
    import threading
    gs = [graphgen.synthetic(4000 + 500 * k, seed=30 + k) for k in range(4)]
    want = []
    for g in gs:
        o = make_oracle(g)
        o.solve(oracle_params(), 90)
        want.append(o.x.copy())
    got, errs = [None] * 4, []

    def work(k):
        try:
            xs = []
            for rep in range(2):
                with GraphRegularizer(gs[k].pos, gs[k].edges, gs[k].alpha, gs[k].beta, gs[k].z,
                                      gs[k].wgt, tile_own=48 if rep else 0) as r:
                    for n in (30, 30, 30):
                        r.step(default_params(), n)
                    xs.append(r.download()[0])
            assert np.array_equal(xs[0].view(np.uint32), xs[1].view(np.uint32))
            got[k] = xs[0]
        except Exception as e:  # noqa
            errs.append((k, repr(e)))

    th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for k in range(4):
        assert_bit_equal(got[k], want[k], "thread %d" % k)


@pytest.mark.parametrize("kind", [0, 1])
def test_graph_filters(gpu, kind):
    """Row a9: median / low-pass graph filter, also followed by the regulariser."""
    g = graphgen.synthetic(5000, seed=8)
    o, r = run_both(g, {}, 0)
    for _ in range(2):
        o.graph_filter(kind)
    r.graph_filter(kind, passes=2)
    x, xb = r.download()[0], r.download_bar()[0]
    assert_bit_equal(x, o.x, "filtered x")
    assert_bit_equal(xb, o.xb, "filtered x_bar")
    assert np.abs(x - g.z).max() > 1e-3  # it did something: the 5 % outliers are pulled in
    o.solve(oracle_params(), 20)
    r.step(default_params(), 20)
    compare_state(o, r, "after filter kind %d" % kind)


def test_update_data_keeps_topology(gpu):
    """flame_hip_graph_update_data: new z / weights / x0 on the same graph == a fresh upload."""
    g = graphgen.synthetic(3000, seed=13)
    rng = np.random.default_rng(14)
    r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt)
    r.step(default_params(), 37)  # dirty state, odd number of ping-pong flips
    for x0 in (None, (g.z + 0.01).astype(np.float32)):
        z2 = (g.z + rng.normal(0, 0.05, g.V)).clip(0.01).astype(np.float32)
        w2 = rng.uniform(0.5, 2.0, g.V).astype(np.float32)
        r.update_data(z2, w2, x0)
        r.step(default_params(), 41)
        o = make_oracle(g)
        o.z[:], o.wgt[:] = z2, w2
        o.set_state(x=z2 if x0 is None else x0, xb=z2 if x0 is None else x0)
        o.solve(oracle_params(), 41)
        compare_state(o, r, "update_data")


@pytest.mark.parametrize("kw", [
    dict(data_factor=0.1, step_x=2e-3, step_q=60.0, theta=0.5),       # cfg comment: "0.1 for lvl5"
    dict(data_factor=0.25, step_x=5e-4, step_q=250.0, theta=0.0),     # "0.25 for lvl3", no extrapolation
    dict(data_factor=0.15, step_x=1e-3, step_q=125.0, theta=1.0, x_min=0.3, x_max=0.9),  # active clamp
])
def test_parameter_and_weight_variants(gpu, kw):
    """Other regulariser parameters (reference cfg/flame_offline_tum.yaml:93-96 comments), adaptive
    data weights (1/var, yaml :89) incl. zero-weight vertices, and x initialised from a prediction
    (init_with_prediction, yaml :91) instead of the data term."""
    from flame_ros_amd.regularizer import default_params as gp
    g = graphgen.synthetic(4000, seed=15)
    rng = np.random.default_rng(16)
    wgt = (1.0 / rng.uniform(1e-3, 1e-2, g.V)).astype(np.float32)
    wgt[::17] = 0.0
    x0 = (g.z + rng.normal(0, 0.03, g.V)).clip(0.01).astype(np.float32)
    for opts in (dict(path=1), dict(path=2), dict(path=2, tile_own=40, tile_depth=3)):
        o = make_oracle(g, x0=x0)
        o.wgt[:] = wgt
        r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, wgt, x0=x0, **opts)
        o.solve(oracle_params(**kw), 77)
        r.step(gp(**kw), 77)
        compare_state(o, r, "%s %s" % (kw, opts))


def test_fuzz_sizes_and_tile_options(gpu):
    """Random graph sizes, tile sizes, halo depths, workgroup sizes and launch splits (incl. tile
    counts that are not multiples of 8 for the XCD-aware block map, tiles with very few own
    vertices, and more tiles than the halo can separate): always the oracle's bits."""
    rng = np.random.default_rng(2026)
    for trial in range(24):
        V = int(rng.integers(3, 2500))
        g = graphgen.synthetic(V, seed=100 + trial) if V >= 4 else None
        if g is None:
            continue
        own = int(rng.integers(4, max(5, V // 2)))
        depth = int(rng.integers(1, 9))
        opts = dict(path=2, tile_own=own, tile_depth=depth, balance=int(rng.integers(0, 2)),
                    order_mode=int(rng.integers(0, 2)), use_graph=int(rng.integers(0, 2)))
        if rng.random() < 0.3:
            opts["tile_threads"] = int(rng.choice([256, 512, 1024]))
        chunks = [int(c) for c in rng.integers(0, 12, size=3)]
        try:
            o, r = run_both(g, opts, None, state_seed=trial, chunks=chunks)
        except Exception as e:  # a forced workgroup size may not fit: then the error must be clean
            from flame_ros_amd.lib import FlameHipError
            assert isinstance(e, FlameHipError) and e.code == -1 and "tile_threads" in opts, (opts, e)
            continue
        compare_state(o, r, "fuzz V=%d %s %s" % (V, opts, chunks))
