"""flame::utils::delaunay (include/flame/utils/delaunay.h, the facade's default FrontEnd::triangulate):
against SciPy's Delaunay on generic points, and by the defining properties -- exact, in Python integers --
on degenerate ones (pixel lattices, collinear runs, duplicates)."""
import os
import subprocess
from fractions import Fraction

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("dt") / "delaunay_test")
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-Werror", "-pthread", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "delaunay_test.cc"), "-o", out])
    return out


def run(exe, pts, threads=1):
    pts = np.asarray(pts, np.float32)
    txt = "%d\n" % len(pts) + "".join("%.9g %.9g\n" % (x, y) for x, y in pts)
    out = subprocess.run([exe, str(threads)], input=txt.encode(), stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
    if out[0] == "FAIL":
        return None
    return np.array([[int(v) for v in l.split()] for l in out if l.strip()], np.int64).reshape(-1, 3)


def snapped(pts):
    """The lattice the header works on: coordinates x 2^16, rounded -- as Python integers."""
    return [(int(round(float(np.float32(x)) * 65536.0)), int(round(float(np.float32(y)) * 65536.0))) for x, y in pts]


def orient(a, b, c):
    return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])


def in_circle(a, b, c, d):
    m = [[p[0] - d[0], p[1] - d[1], (p[0] - d[0]) ** 2 + (p[1] - d[1]) ** 2] for p in (a, b, c)]
    return (m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]) - m[1][2] * (m[0][0] * m[2][1] - m[0][1] * m[2][0])
            + m[2][2] * (m[0][0] * m[1][1] - m[0][1] * m[1][0]))


def check_properties(pts, tris):
    """Counter-clockwise triangles; every edge in one triangle (hull) or two; Euler: T = 2n - 2 - h for the n
    DISTINCT points with h of them on the hull; no vertex strictly inside a circumcircle (locally: the apex
    across every interior edge -- for a triangulation of a convex region that is the global property)."""
    P = snapped(pts)
    used = sorted(set(tris.ravel().tolist()))
    distinct = {}
    for i, q in enumerate(P):
        distinct.setdefault(q, i)
    assert sorted(distinct.values()) == used, "every distinct point is a vertex, every copy but the first is not"
    edges = {}
    for t in tris:
        a, b, c = (int(v) for v in t)
        assert orient(P[a], P[b], P[c]) > 0
        for u, w, x in ((a, b, c), (b, c, a), (c, a, b)):
            assert (u, w) not in edges, "an oriented edge belongs to one triangle"
            edges[(u, w)] = x
    hull = 0
    for (u, w), x in edges.items():
        if (w, u) not in edges:
            hull += 1
            continue
        y = edges[(w, u)]
        assert in_circle(P[u], P[w], P[x], P[y]) <= 0, (u, w, x, y)
    assert len(tris) == 2 * len(used) - 2 - hull
    # hull edges form the convex hull: every point is on or left of each of them
    for (u, w), x in edges.items():
        if (w, u) not in edges:
            assert all(orient(P[u], P[w], P[i]) >= 0 for i in used)


def canon(tris):
    t = np.asarray(tris, np.int64)
    k = np.argmin(t, 1)
    t = np.stack([t[np.arange(len(t)), (k + j) % 3] for j in range(3)], 1)
    return t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))]


@pytest.mark.parametrize("n,seed", [(3, 0), (4, 1), (10, 2), (200, 3), (5000, 4)])
def test_matches_scipy_on_generic_points(exe, n, seed):
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    pts = (rng.random((n, 2)) * np.array([640.0, 480.0]) + 128.0).astype(np.float32)  # (>= 128: on the lattice)
    got = run(exe, pts)
    assert got is not None
    check_properties(pts, got) if n <= 200 else None
    ref = Delaunay(pts.astype(np.float64)).simplices
    # same orientation convention before comparing: counter-clockwise in (x, y)
    P = pts.astype(np.float64)
    d = (P[ref[:, 1], 0] - P[ref[:, 0], 0]) * (P[ref[:, 2], 1] - P[ref[:, 0], 1]) - \
        (P[ref[:, 1], 1] - P[ref[:, 0], 1]) * (P[ref[:, 2], 0] - P[ref[:, 0], 0])
    ref = np.where(d[:, None] > 0, ref, ref[:, [0, 2, 1]])
    assert np.array_equal(canon(got), canon(ref))


def test_degenerate_inputs(exe):
    # a pixel lattice: every cell is cocircular
    ix, iy = np.meshgrid(np.arange(12), np.arange(9))
    lattice = np.stack([ix.ravel() * 16.0 + 8.0, iy.ravel() * 16.0 + 8.0], 1)
    check_properties(lattice, run(exe, lattice))
    # one feature per 16-pixel cell at integer pixels (what the detector produces)
    rng = np.random.default_rng(5)
    cells = np.stack([ix.ravel() * 16 + rng.integers(0, 16, ix.size), iy.ravel() * 16 + rng.integers(0, 16, ix.size)], 1)
    check_properties(cells, run(exe, cells))
    # collinear runs inside the set, duplicates, tiny coordinates (off the lattice: snapped)
    pts = np.array([[0, 0], [1, 0], [2, 0], [3, 0], [0, 1], [3, 1], [1.5, 0.25], [1.5, 0.25], [0, 0], [2.5, 1e-3]], np.float32)
    check_properties(pts, run(exe, pts))
    # nothing to triangulate
    assert run(exe, np.array([[0, 0], [1, 1]], np.float32)) is None
    assert run(exe, np.array([[0, 0], [1, 1], [2, 2], [5, 5]], np.float32)) is None
    assert run(exe, np.array([[0, 0], [1, 0], [np.inf, 3]], np.float32)) is None
    assert run(exe, np.zeros((0, 2), np.float32)) is None


def test_small_integer_sets_stress(exe):
    """Many tiny point sets on a coarse integer grid: ties in both cut directions, collinear subsets,
    cocircular quadruples and duplicates at every level of the recursion."""
    rng = np.random.default_rng(7)
    for trial in range(150):
        n = int(rng.integers(3, 40))
        pts = rng.integers(0, 7, (n, 2)).astype(np.float32) * np.float32(8.0) + np.float32(128.0)
        tris = run(exe, pts)
        P = set(map(tuple, pts.tolist()))
        xs, ys = {p[0] for p in P}, {p[1] for p in P}
        collinear = len(P) < 3 or all(orient(*(snapped([a, b, c]))) == 0
                                      for a in list(P)[:1] for b in list(P)[1:2] for c in P)
        if collinear:
            assert tris is None, trial
            continue
        assert tris is not None, trial
        check_properties(pts, tris)
    # a circle through 12 lattice points, a vertical and a horizontal line through its centre
    circle = [(5, 0), (4, 3), (3, 4), (0, 5), (-3, 4), (-4, 3), (-5, 0), (-4, -3), (-3, -4), (0, -5), (3, -4), (4, -3)]
    pts = np.array(circle + [(0, k) for k in range(-4, 5)] + [(k, 0) for k in range(-4, 5) if k], np.float32) * 4 + 200
    check_properties(pts, run(exe, pts))


def test_threads_give_the_same_triangulation(exe):
    """threads = 2 / 4: the top levels of the recursion on threads of their own, parts joined afterwards --
    the same triangles (the predicates are exact, the result is unique for generic points), also when the
    input is a lattice (any valid choice inside cocircular cells: checked by the properties)."""
    rng = np.random.default_rng(9)
    pts = (rng.random((6000, 2)) * np.array([640.0, 480.0]) + 128.0).astype(np.float32)
    ref = canon(run(exe, pts, 1))
    for th in (2, 4, 8, 16, 23):
        assert np.array_equal(canon(run(exe, pts, th)), ref), th
    ix, iy = np.meshgrid(np.arange(80), np.arange(60))
    lattice = np.stack([ix.ravel() * 8.0 + 130.0, iy.ravel() * 8.0 + 130.0], 1)
    check_properties(lattice, run(exe, lattice, 4))
    check_properties(lattice, run(exe, lattice, 16))
    # duplicates, a collinear run and clusters on 16 threads (the bucket sort and the cuts see ties)
    rng = np.random.default_rng(4)
    clus = np.concatenate([rng.normal((300, 300), 3, (3000, 2)), np.stack([np.arange(1500) * 0.25 + 130, np.full(1500, 222.0)], 1),
                           rng.integers(130, 400, (3000, 2)).astype(np.float64)]).astype(np.float32)
    check_properties(clus, run(exe, clus, 16))


def test_sanitizers_clean(tmp_path):
    """The triangulator under AddressSanitizer + UBSan (serial and threaded, generic and degenerate input)."""
    out = str(tmp_path / "delaunay_asan")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-pthread", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "delaunay_test.cc"), "-o", out])
    rng = np.random.default_rng(3)
    ix, iy = np.meshgrid(np.arange(70), np.arange(60))
    cases = [((rng.random((3000, 2)) * np.array([640.0, 480.0])).astype(np.float32), 1),
             ((rng.random((6000, 2)) * np.array([640.0, 480.0])).astype(np.float32), 4),
             (np.stack([ix.ravel() * 8.0 + 130.0, iy.ravel() * 8.0 + 130.0], 1).astype(np.float32), 4)]
    cases += [(rng.integers(0, 6, (int(rng.integers(3, 50)), 2)).astype(np.float32) * 4 + 130, 1) for _ in range(10)]
    for pts, th in cases:
        txt = "%d\n" % len(pts) + "".join("%.9g %.9g\n" % (x, y) for x, y in pts)
        p = subprocess.run([out, str(th)], input=txt.encode(), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()[-2000:]


def test_thread_sanitizer_clean(tmp_path):
    """The pool, the shared quad-edge array with per-subtree ranges and the level-by-level cuts / merges under
    ThreadSanitizer: no data race on 4, 8 and 16 threads (random points, a lattice)."""
    out = str(tmp_path / "delaunay_tsan")
    r = subprocess.run(["g++", "-std=c++11", "-O1", "-g", "-fsanitize=thread", "-pthread", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "cpp", "delaunay_test.cc"), "-o", out], capture_output=True)
    if r.returncode != 0:
        pytest.skip("no ThreadSanitizer runtime in this toolchain")
    rng = np.random.default_rng(5)
    ix, iy = np.meshgrid(np.arange(70), np.arange(60))
    cases = [((rng.random((1500, 2)) * np.array([640.0, 480.0])).astype(np.float32), 4),
             ((rng.random((9000, 2)) * np.array([640.0, 480.0])).astype(np.float32), 8),
             ((rng.random((9000, 2)) * np.array([640.0, 480.0])).astype(np.float32), 16),
             (np.stack([ix.ravel() * 8.0 + 130.0, iy.ravel() * 8.0 + 130.0], 1).astype(np.float32), 8)]
    for pts, th in cases:
        txt = "%d\n" % len(pts) + "".join("%.9g %.9g\n" % (x, y) for x, y in pts)
        p = subprocess.run([out, str(th)], input=txt.encode(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert p.returncode == 0 and b"ThreadSanitizer" not in p.stderr, p.stderr.decode()[-3000:]
