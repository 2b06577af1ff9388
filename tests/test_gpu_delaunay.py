"""-m gpu: flame_hip_delaunay (flame_ros_amd/csrc/delaunay_dev.hip) -- SURVEY.md 8 row f3's first leg, the Delaunay
triangulation of a frame's features, on the GPU: against SciPy on generic points (the triangulation is unique there), and
by the defining properties -- exact, in Python integers -- on degenerate ones (pixel lattices, cocircular rings, collinear
runs, duplicates, clusters), exactly the cases tests/test_delaunay.py holds for the host triangulator of the same contract.
(The reference has no test of its own for this step: upstream calls Shewchuk's Triangle; stat key `triangulate`,
/root/reference/msg/FlameStats.msg:44.)"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_delaunay import canon, check_properties, orient, snapped

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def handle(gpu):
    from flame_ros_amd.regularizer import GraphRegularizer
    h = GraphRegularizer.empty()
    yield h
    h.close()


def scipy_ccw(pts):
    from scipy.spatial import Delaunay
    P = pts.astype(np.float64)
    ref = Delaunay(P).simplices
    d = (P[ref[:, 1], 0] - P[ref[:, 0], 0]) * (P[ref[:, 2], 1] - P[ref[:, 0], 1]) - \
        (P[ref[:, 1], 1] - P[ref[:, 0], 1]) * (P[ref[:, 2], 0] - P[ref[:, 0], 0])
    return np.where(d[:, None] > 0, ref, ref[:, [0, 2, 1]])


@pytest.mark.parametrize("n,seed", [(3, 0), (4, 1), (10, 2), (200, 3), (1200, 6), (5000, 4), (10000, 5), (50000, 7), (200000, 8)])
def test_matches_scipy_on_generic_points(handle, n, seed):
    rng = np.random.default_rng(seed)
    pts = (rng.random((n, 2)) * np.array([640.0, 480.0]) + 128.0).astype(np.float32)  # (>= 128: on the lattice)
    got = handle.delaunay(pts)
    if n <= 200:
        check_properties(pts, got)
    assert np.array_equal(canon(got), canon(scipy_ccw(pts)))
    # the list itself: every triangle starts at its smallest vertex, the list is ordered by it, and a second call
    # returns the same list (not just the same set)
    assert np.all(got[:, 0] < got[:, 1]) and np.all(got[:, 0] < got[:, 2]) and np.all(np.diff(got[:, 0]) >= 0)
    assert np.array_equal(got, handle.delaunay(pts))
    assert handle.info("delaunay_live") == n and 2 * n - 2 - handle.info("delaunay_hull") == len(got)


def test_clustered_and_skewed_distributions(handle):
    """Nothing about the grid assumes uniform features: clusters (cells with hundreds of points), a thin strip (one grid
    row), points on a few image rows, a huge aspect ratio."""
    rng = np.random.default_rng(11)
    clus = np.concatenate([rng.normal((300, 300), 3, (3000, 2)), rng.normal((500, 200), 0.5, (2000, 2)),
                           rng.random((500, 2)) * np.array([640.0, 480.0]) + 128.0]).astype(np.float32)
    strip = np.stack([rng.random(4000) * 600 + 130, rng.random(4000) * 0.5 + 300], 1).astype(np.float32)
    wide = np.stack([rng.random(3000) * 8000 - 4000, rng.random(3000) * 3 - 1], 1).astype(np.float32)
    for pts in (clus, strip, wide):
        pts = np.unique(pts, axis=0)
        rng.shuffle(pts)
        got = handle.delaunay(pts)
        want = scipy_ccw((np.round(pts.astype(np.float64) * 65536.0)))  # (what is triangulated: the snapped points)
        if np.array_equal(canon(got), canon(want)):
            continue
        check_properties(pts, got)  # (snapping may have made cocircular / collinear sets: any valid choice)


def test_degenerate_inputs(handle):
    # a pixel lattice: every cell is cocircular
    ix, iy = np.meshgrid(np.arange(12), np.arange(9))
    lattice = np.stack([ix.ravel() * 16.0 + 8.0, iy.ravel() * 16.0 + 8.0], 1)
    check_properties(lattice, handle.delaunay(lattice))
    # one feature per 16-pixel cell at integer pixels (what the detector produces)
    rng = np.random.default_rng(5)
    cells = np.stack([ix.ravel() * 16 + rng.integers(0, 16, ix.size), iy.ravel() * 16 + rng.integers(0, 16, ix.size)], 1)
    check_properties(cells, handle.delaunay(cells))
    # collinear runs inside the set, duplicates, tiny coordinates (off the lattice: snapped)
    pts = np.array([[0, 0], [1, 0], [2, 0], [3, 0], [0, 1], [3, 1], [1.5, 0.25], [1.5, 0.25], [0, 0], [2.5, 1e-3]], np.float32)
    check_properties(pts, handle.delaunay(pts))
    # nothing to triangulate: an empty list (the facade reports the frame as failed, like the host triangulator's `false`)
    assert len(handle.delaunay(np.array([[0, 0], [1, 1]], np.float32))) == 0
    assert len(handle.delaunay(np.array([[0, 0], [1, 1], [2, 2], [5, 5]], np.float32))) == 0
    assert len(handle.delaunay(np.array([[3, 3], [3, 3], [3, 3], [3, 3]], np.float32))) == 0
    assert len(handle.delaunay(np.zeros((0, 2), np.float32))) == 0
    from flame_ros_amd.lib import FlameHipError
    with pytest.raises(FlameHipError):
        handle.delaunay(np.array([[0, 0], [1, 0], [np.inf, 3]], np.float32))
    with pytest.raises(FlameHipError):
        handle.delaunay(np.array([[0, 0], [1, 0], [9000.0, 3]], np.float32))  # outside |u| < 2^13


def test_small_integer_sets_stress(handle):
    """Many tiny point sets on a coarse integer grid: collinear subsets, cocircular quadruples and larger rings,
    duplicates -- every star has to make the same choice inside every cocircular polygon."""
    rng = np.random.default_rng(7)
    for trial in range(300):
        n = int(rng.integers(3, 60))
        pts = rng.integers(0, 7, (n, 2)).astype(np.float32) * np.float32(8.0) + np.float32(128.0)
        tris = handle.delaunay(pts)
        P = list(set(map(tuple, pts.tolist())))
        collinear = len(P) < 3 or all(orient(*(snapped([P[0], P[1], c]))) == 0 for c in P)
        if collinear:
            assert len(tris) == 0, trial
            continue
        check_properties(pts, tris)
    # a circle through 12 lattice points, a vertical and a horizontal line through its centre
    circle = [(5, 0), (4, 3), (3, 4), (0, 5), (-3, 4), (-4, 3), (-5, 0), (-4, -3), (-3, -4), (0, -5), (3, -4), (4, -3)]
    pts = np.array(circle + [(0, k) for k in range(-4, 5)] + [(k, 0) for k in range(-4, 5) if k], np.float32) * 4 + 200
    check_properties(pts, handle.delaunay(pts))
    # the bare ring (one polygon of 12 cocircular points: a fan from vertex 0), in every rotation of the ids
    ring = np.array(circle, np.float32) * 4 + 200
    for k in range(12):
        r = np.roll(ring, k, axis=0)
        tris = handle.delaunay(r)
        check_properties(r, tris)
        assert len(tris) == 10 and np.all(tris[:, 0] == 0)


def test_lattices_at_frame_size(handle):
    """A full 80 x 60 lattice (4 800 cocircular cells) and integer-pixel features at the BASELINE sizes."""
    ix, iy = np.meshgrid(np.arange(80), np.arange(60))
    lattice = np.stack([ix.ravel() * 8.0 + 130.0, iy.ravel() * 8.0 + 130.0], 1)
    check_properties(lattice, handle.delaunay(lattice))
    rng = np.random.default_rng(4)
    clus = np.concatenate([rng.normal((300, 300), 3, (3000, 2)), np.stack([np.arange(1500) * 0.25 + 130, np.full(1500, 222.0)], 1),
                           rng.integers(130, 400, (3000, 2)).astype(np.float64)]).astype(np.float32)
    check_properties(clus, handle.delaunay(clus))
    pix = np.stack([rng.integers(0, 640, 10000), rng.integers(0, 480, 10000)], 1).astype(np.float32)  # duplicates, ties
    check_properties(pix, handle.delaunay(pix))


@pytest.mark.parametrize("workload", ["tum", "5k", "euroc"])
def test_facade_frames_do_not_depend_on_the_triangulator(gpu, workload):
    """flame::Flame::update() from features alone (tools/facade_bench.cc with a FrontEnd that only tracks), the built-in
    triangulation once on the GPU (flame_hip_delaunay, Params::triangulate_on_gpu = true, the default) and once on the host
    (include/flame/utils/delaunay.h): generic features have ONE Delaunay triangulation, so the frame's edge list and every
    bit of its regularised idepths are the same (the triangle lists differ in order only) -- and equal to the frame handed
    over with the triangulation the graph generator made (SciPy's)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools import facade_bench
    on_gpu = facade_bench.run(workload, repeats=1, getters=1, env={"FLAME_BENCH_FRONTEND": "1", "FLAME_BENCH_TRI_GPU": "1"})
    on_host = facade_bench.run(workload, repeats=1, getters=1, env={"FLAME_BENCH_FRONTEND": "1", "FLAME_BENCH_TRI_GPU": "0"})
    given = facade_bench.run(workload, repeats=1, getters=1)
    assert on_gpu["T"] == on_host["T"] == given["T"] and on_gpu["E"] == on_host["E"] == given["E"]
    assert on_gpu["x_hash"] == on_host["x_hash"] == given["x_hash"], (on_gpu["x_hash"], on_host["x_hash"], given["x_hash"])
    assert on_gpu["triangulate_ms_p50"] > 0 and on_host["triangulate_ms_p50"] > 0


def test_random_sets_fuzz(handle):
    """150 random sets of 3 .. 3 000 generic points in boxes of random size and aspect (grids of 1 x 1 to 39 x 39 cells, open and
    closed stars in every proportion): the unique Delaunay triangulation, as SciPy finds it."""
    rng = np.random.default_rng(123)
    for trial in range(150):
        n = int(rng.integers(3, 3000)) if trial % 3 else int(rng.integers(3, 40))
        w, h = float(rng.uniform(2, 4000)), float(rng.uniform(2, 4000))
        x0, y0 = float(rng.uniform(-4000, 4000 - w)), float(rng.uniform(-4000, 4000 - h))
        pts = (rng.random((n, 2)) * np.array([w, h]) + np.array([x0, y0])).astype(np.float32)
        got = handle.delaunay(pts)
        snapped_pts = np.round(pts.astype(np.float64) * 65536.0)
        if len(np.unique(snapped_pts, axis=0)) != n:
            continue  # (a coincidence after snapping: covered by the degenerate cases)
        want = scipy_ccw(snapped_pts)
        if not np.array_equal(canon(got), canon(want)):
            check_properties(pts, got)  # (nearly cocircular / collinear for floating point: any exact answer)
            assert len(got) == len(want), trial


def test_two_handles_triangulate_concurrently(gpu):
    """Two handles, two host threads, 60 frames each (sizes on both sides of the small-frame path), the calls interleaving
    on the device: every list equals the one a quiet handle returns."""
    import threading
    from flame_ros_amd.regularizer import GraphRegularizer
    rng = np.random.default_rng(77)
    frames = [(rng.random((n, 2)) * np.array([640.0, 480.0])).astype(np.float32) for n in (1200, 5000, 300, 2048, 2049, 10000)]
    with GraphRegularizer.empty() as h0:
        want = [h0.delaunay(f) for f in frames]
    bad = []

    def stream(tid):
        with GraphRegularizer.empty() as h:
            for k in range(60):
                i = (k + tid) % len(frames)
                if not np.array_equal(h.delaunay(frames[i]), want[i]):
                    bad.append((tid, k))
    th = [threading.Thread(target=stream, args=(t,)) for t in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    assert not bad, bad[:5]


def test_near_cocircular_and_near_collinear_sets(handle):
    """What the floating-point filters must hand to the exact arithmetic: points within a few lattice units of a common
    circle (every in-circle test of a star is nearly zero), of a common line, and both at once -- checked by the defining
    properties in exact integer arithmetic."""
    rng = np.random.default_rng(31)
    for trial in range(40):
        n = int(rng.integers(8, 120))
        th = np.sort(rng.random(n) * 2 * np.pi)
        R = float(rng.uniform(20, 2000))
        c = rng.uniform(-1000, 1000, 2)
        ring = np.stack([np.cos(th), np.sin(th)], 1) * R + c
        jitter = rng.integers(-2, 3, (n, 2)) / 65536.0 * (trial % 3)       # 0 .. 2 lattice units
        line = np.stack([np.linspace(-R, R, n), np.linspace(-R, R, n) * float(rng.uniform(-2, 2))], 1) + c
        line += rng.integers(-1, 2, (n, 2)) / 65536.0 * (trial % 2)
        inner = rng.uniform(-0.5, 0.5, (n // 4, 2)) * R + c
        pts = np.concatenate([ring + jitter, line, inner]).astype(np.float64)
        pts = np.clip(pts, -8000, 8000).astype(np.float32)
        tris = handle.delaunay(pts)
        assert len(tris) > 0
        check_properties(pts, tris)


@pytest.mark.parametrize("n", [1200, 10000])
def test_graph_sync_on_the_library_s_own_list(gpu, n):
    """flame_hip_graph_sync with tris = NULL and the T of the last flame_hip_delaunay on the handle reads the list where
    the library still holds it (page-locked): the same graph, the same bits after 50 iterations as with the list handed
    over -- and a T / V that does not match that list is refused."""
    from flame_ros_amd.lib import FlameHipError
    from flame_ros_amd.regularizer import GraphRegularizer, default_params, default_sync_params
    rng = np.random.default_rng(n)
    pts = (rng.random((n, 2)) * np.array([640.0, 480.0])).astype(np.float32)
    mu = (0.5 + 0.001 * pts[:, 0] + 0.05 * rng.standard_normal(n)).astype(np.float32)
    var = np.full(n, 1e-4, np.float32)
    sp = default_sync_params()
    out = []
    for own_list in (False, True):
        with GraphRegularizer.empty() as h:
            tris = h.delaunay(pts)
            h.sync_features(pts, mu, var, len(tris) if own_list else tris, sp)
            edges = h.edges().copy()
            h.step(default_params(), 50)
            out.append((edges, h.download()[0].copy()))
            if own_list:
                with pytest.raises(FlameHipError):
                    h.sync_features(pts, mu, var, len(tris) - 1, sp)
                with pytest.raises(FlameHipError):
                    h.sync_features(pts[:-1], mu[:-1], var[:-1], len(tris), sp)
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32))


@pytest.mark.parametrize("n", [600, 1200, 10000, 50000])
def test_keep_mode_list_stays_on_the_device(gpu, n):
    """r05: flame_hip_delaunay with tri_cap = 0 / tris = NULL returns only T; the graph sync that
    follows reads the list on the DEVICE (one-launch plan at 600 / 1.2 k, the staged build above; 600 vertices: the host
    builder, which has to wait for the list's host copy), and flame_hip_delaunay_list hands the same triangles out afterwards.
    Same edges, same bits after 50 iterations as with the list handed over by the caller."""
    from flame_ros_amd.regularizer import GraphRegularizer, default_params, default_sync_params
    rng = np.random.default_rng(n + 5)
    pts = (rng.random((n, 2)) * np.array([640.0, 480.0])).astype(np.float32)
    mu = (0.5 + 0.001 * pts[:, 0] + 0.05 * rng.standard_normal(n)).astype(np.float32)
    var = np.full(n, 1e-4, np.float32)
    sp = default_sync_params()
    with GraphRegularizer.empty(tile_single_max=640) as ref, GraphRegularizer.empty(tile_single_max=640) as h:
        tris = ref.delaunay(pts)
        ref.sync_features(pts, mu, var, tris, sp)
        ref.step(default_params(), 50)
        for frame in range(3):  # (a stream: the second and third frame reuse the partition / the one-launch plan)
            T = h.delaunay_keep(pts)
            assert T == len(tris)
            h.sync_features(pts, mu, var, T, sp)
            h.step(default_params(), 50, sync=False)
            got = h.delaunay_list()  # (while the GPU iterates)
            assert np.array_equal(got, tris), frame
            assert np.array_equal(h.edges(), ref.edges())
            assert np.array_equal(h.download()[0].view(np.uint32), ref.download()[0].view(np.uint32)), frame
