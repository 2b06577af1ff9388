"""-m gpu: row f3 -- the plan built ON THE GPU (csrc/plan_dev.hip) equals the host builder's plan
(csrc/plan.cpp) array for array: vertex / edge permutations, incidence CSR, triangle CSR, tile
descriptors, gather lists, local edge records, incidence slots.  Also on a frame STREAM (the second
frame balances in one pass from the integer cost-density grid kept by each builder)."""
import ctypes as C

import numpy as np
import pytest

from flame_ros_amd import lib as _l
from flame_ros_amd.regularizer import GraphRegularizer, default_params
from tests.util import assert_bit_equal, graphgen, host_reference_opts, make_oracle, oracle_params

pytestmark = pytest.mark.gpu

ARRAYS = [("v_i2o", np.int32), ("v_o2i", np.int32), ("e_i2o", np.int32), ("e_o2i", np.int32),
          ("grow", np.int32), ("ginc", np.int32), ("eij", np.int32), ("ew", np.float32),
          ("tris", np.int32), ("trow", np.int32), ("tinc", np.int32), ("tiles", np.int32),
          ("t_vmap", np.int32), ("t_emap", np.int32), ("t_eij", np.uint32), ("t_ew", np.float32),
          ("t_srow", np.uint32)]
INFO = ["path", "num_tiles", "tile_threads", "tile_ept", "tile_vpt", "tile_depth", "tile_lds_bytes", "tile_slot12"]


def compare_plans(host, dev, what):
    assert dev.info("plan_on_device") == 1, what
    assert host.info("plan_on_device") == 0, what
    for k in INFO:
        assert host.info(k) == dev.info(k), (what, k, host.info(k), dev.info(k))
    for name, dt in ARRAYS:
        a, b = host.plan_array(name, dt), dev.plan_array(name, dt)
        assert a.shape == b.shape, (what, name, a.shape, b.shape)
        if not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
            bad = np.flatnonzero(a.view(np.uint32) != b.view(np.uint32))
            raise AssertionError("%s: plan array %s differs at %d of %d words, first %s" % (
                what, name, len(bad), a.size, bad[:8].tolist()))


CASES = [
    ("5k", dict()),
    ("5k", dict(lane_order=2)),
    ("50k", dict(lane_order=2)),
    ("5k", dict(balance=0)),
    ("5k", dict(tile_own=64, tile_depth=2)),
    ("5k", dict(tile_own=300, tile_depth=6)),
    ("tum", dict()),
    ("euroc", dict()),
    ("50k", dict()),
    ("50k", dict(tile_own=100, tile_depth=3)),
    ("200k", dict()),       # (r05: fat resident tiles, 12-byte slots, depth 2)
    ("v100000", dict()),    # (fat tiles, 16-byte slots, depth 3)
    ("v160000", dict()),    # (fat tiles, 16-byte slots at depth 1 after two deeper attempts did not fit)
    ("v220000", dict()),    # (fat tiles, 12-byte slots at depth 1)
    ("v100000", dict(persist=0)),  # (not resident: two rounds of smaller tiles, as in r04)
]


@pytest.mark.parametrize("name,opts", CASES)
def test_device_plan_equals_host_plan(gpu, name, opts):
    g, _ = graphgen.named(name)
    host = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=-1, **{**host_reference_opts(), **opts})
    dev = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, **opts)
    compare_plans(host, dev, "%s %s" % (name, opts))
    host.close(); dev.close()


@pytest.mark.parametrize("name,plan_device", [("5k", 1), ("5k", 0), ("tum", 0), ("50k", 1)])
def test_lane_order_on_second_solve_equals_build_time(gpu, name, plan_device):
    """lane_order = 1 (default): the plan keeps the sorted lane order until it is solved a SECOND
    time (a frame stream never pays for it); what the device then writes is exactly what
    lane_order = 2 builds (host builder and device builder), and the results stay bit-exact."""
    g, _ = graphgen.named(name)
    kw = dict(plan_device=plan_device)
    if name == "tum": kw["tile_single_max"] = 2048  # one isolated tile
    built = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, lane_order=2, **kw)
    lazy = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, **kw)
    sorted_ = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, lane_order=0, **kw)
    lists = [("t_emap", np.int32), ("t_eij", np.uint32), ("t_ew", np.uint32)]
    for n, dt in lists:
        assert np.array_equal(lazy.plan_array(n, dt), sorted_.plan_array(n, dt)), n
    o = make_oracle(g)
    for k in range(3):
        o.solve(oracle_params(), 21)
        for r in (built, lazy, sorted_):
            r.step(default_params(), 21)
        if k == 0:  # one solve: still the sorted order
            assert np.array_equal(lazy.plan_array("t_emap", np.int32), sorted_.plan_array("t_emap", np.int32))
    differs = False
    for n, dt in lists:
        assert np.array_equal(lazy.plan_array(n, dt), built.plan_array(n, dt)), n
        differs |= not np.array_equal(lazy.plan_array(n, dt), sorted_.plan_array(n, dt))
    assert differs
    for r in (built, lazy, sorted_):
        x, w1, w2, q = r.download()
        assert_bit_equal(x, o.x, "x"); assert_bit_equal(q, o.q, "q")
        r.close()


def _irregular(kind):
    g = graphgen.synthetic(3000, seed=77)
    rng = np.random.default_rng(5)
    edges, alpha, beta = g.edges, g.alpha, g.beta
    if kind == "isolated":      # 30 % of the vertices lose every edge (and their triangles)
        dead = rng.random(g.V) < 0.3
        keep = ~(dead[edges[:, 0]] | dead[edges[:, 1]])
        edges, alpha, beta = edges[keep], alpha[keep], beta[keep]
        tris = g.tris[~dead[g.tris].any(1)]
    elif kind == "multi":       # 5 % of the edges twice, the copy with flipped orientation
        dup = rng.random(len(edges)) < 0.05
        edges = np.concatenate([edges, edges[dup][:, ::-1]])
        alpha = np.concatenate([alpha, alpha[dup]]); beta = np.concatenate([beta, beta[dup]])
        tris = g.tris
    elif kind == "hubs":        # a few vertices of degree ~45: rows long enough for the heap sort of
        hubs = rng.choice(g.V, 6, replace=False)  # the counting CSR; their triangles fan out likewise
        extra, etris = [], []
        for h in hubs:
            d2 = ((g.pos - g.pos[h]) ** 2).sum(1)
            near = np.argsort(d2)[1:60]
            have = set(edges[edges[:, 0] == h, 1].tolist()) | set(edges[edges[:, 1] == h, 0].tolist())
            new = [int(v) for v in near if int(v) not in have][:38]
            extra += [(int(h), v) if k % 2 else (v, int(h)) for k, v in enumerate(new)]
            etris += [(int(h), new[k], new[k + 1]) for k in range(len(new) - 1)]
        extra = np.array(extra, np.int32)
        d = g.pos[extra[:, 0]] - g.pos[extra[:, 1]]
        a = (np.float32(1.0) / np.sqrt((d.astype(np.float32) ** 2).sum(1, dtype=np.float32))).astype(np.float32)
        edges = np.concatenate([edges, extra]); alpha = np.concatenate([alpha, a]); beta = np.concatenate([beta, a])
        tris = np.concatenate([g.tris, np.array(etris, np.int32)])
    else:                       # no edges at all
        edges, alpha, beta = edges[:0], alpha[:0], beta[:0]
        tris = None
    return g, np.ascontiguousarray(edges), alpha, beta, tris


@pytest.mark.parametrize("kind", ["isolated", "multi", "hubs", "noedges"])
def test_device_plan_irregular_graphs(gpu, kind):
    g, edges, alpha, beta, tris = _irregular(kind)
    opts = dict(tile_own=64, tile_depth=3)
    host = GraphRegularizer(g.pos, edges, alpha, beta, g.z, g.wgt, tris=tris, device=-1, **{**host_reference_opts(), **opts})
    dev = GraphRegularizer(g.pos, edges, alpha, beta, g.z, g.wgt, tris=tris, device=0, **opts)
    compare_plans(host, dev, kind)
    if kind == "hubs":
        grow = host.plan_array("grow", np.int32)
        assert np.diff(grow).max() > 24  # long rows are there
    from oracle import COracle
    o = COracle(g.pos, edges, alpha, beta, g.z, g.wgt)
    o.solve(oracle_params(), 25)
    dev.step(default_params(), 25)
    x, w1, w2, q = dev.download()
    assert_bit_equal(x, o.x, kind + " x")
    assert_bit_equal(q, o.q, kind + " q")
    host.close(); dev.close()


def test_star_graph_falls_back_to_global_path(gpu):
    """One hub of degree V-1: no tile plan can hold it (its incidence row alone exceeds the LDS).  Both
    builders give up on tiles -- the device builder's counting CSR sorts the 6 k-entry row by heap
    sort, it does not hang -- and the global path matches the oracle."""
    g = graphgen.synthetic(6000, seed=3)
    hub = 17
    others = np.array([v for v in range(g.V) if v != hub], np.int32)
    edges = np.stack([np.full(len(others), hub, np.int32), others], 1)
    edges[::2] = edges[::2, ::-1]
    edges = np.ascontiguousarray(edges)
    d = g.pos[edges[:, 0]] - g.pos[edges[:, 1]]
    a = (np.float32(1) / np.sqrt((d ** 2).sum(1, dtype=np.float32))).astype(np.float32)
    from oracle import COracle
    o = COracle(g.pos, edges, a, a, g.z, g.wgt)
    o.solve(oracle_params(), 40)
    for plan_device in (1, 0):
        r = GraphRegularizer(g.pos, edges, a, a, g.z, g.wgt, tris=None, plan_device=plan_device)
        assert r.info("path") == _l.PATH_GLOBAL
        r.step(default_params(), 40)
        assert_bit_equal(r.download()[0], o.x, "star x")
        r.close()


def test_device_plan_frame_stream(gpu):
    """Consecutive frames of different size on ONE handle each: from the second frame on both
    builders balance in one pass from their cost-density grid; every frame's plan is identical and
    the solve on the device-built plan is bit-exact against the oracle."""
    host = dev = None
    for k, V in enumerate((20000, 20600, 19500, 20000, 5000, 5100)):
        g = graphgen.synthetic(V, seed=40 + k)
        if host is None:
            host = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=-1, **host_reference_opts())
            # (plan_reuse=0: the partition-reuse shortcut of frame streams is the device builder's own;
            # the array-for-array comparison is about the exact bisection both builders implement)
            dev = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, plan_reuse=0)
        else:
            host.reupload(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris)
            dev.reupload(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris)
        compare_plans(host, dev, "frame %d (V=%d)" % (k, V))
        o = make_oracle(g)
        o.solve(oracle_params(), 33)
        dev.step(default_params(), 33)
        x, w1, w2, q = dev.download()
        assert_bit_equal(x, o.x, "frame %d x" % k)
        assert_bit_equal(q, o.q, "frame %d q" % k)
    host.close(); dev.close()


def test_partition_reuse_on_a_frame_stream(gpu):
    """Frame streams take the partition of a frame from the previous frame's tile map (plan_reuse, the
    default): frames of similar size are not bisected again, the plan is valid (the solve is the
    oracle's bits, as on any partition), a frame the map does not suit (scene change: all features in
    one corner; a much larger frame) is rebuilt by exact bisection."""
    dev = None
    rng = np.random.default_rng(3)
    sizes = [(20000, None), (20400, None), (19000, None), (20000, "corner"), (20100, None), (20300, None),
             (9000, None), (9300, None)]
    reused = []
    for k, (V, kind) in enumerate(sizes):
        g = graphgen.synthetic(V, seed=70 + k)
        if kind == "corner":  # every feature inside the top-left eighth of the image
            pos = (rng.random((V, 2)) * np.array([80.0, 60.0])).astype(np.float32)
            g = graphgen.from_points(pos, 640, 480, np.random.Generator(np.random.PCG64(5)))
        if dev is None:
            dev = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0)
        else:
            dev.reupload(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris)
        assert dev.info("plan_on_device") == 1 and dev.info("path") == _l.PATH_TILE
        reused.append(dev.info("plan_reused"))
        o = make_oracle(g)
        o.solve(oracle_params(), 25)
        dev.step(default_params(), 25)
        x, w1, w2, q = dev.download()
        assert_bit_equal(x, o.x, "frame %d x" % k)
        assert_bit_equal(q, o.q, "frame %d q" % k)
        # the tiles cover the graph exactly once
        td = dev.plan_array("tiles", np.int32).reshape(dev.info("num_tiles"), -1)
        assert td[:, 1].sum() == g.V and td[:, 1].min() >= 1
    #          first similar similar corner back(sits out one frame) similar smaller similar
    assert reused == [0, 1, 1, 0, 0, 1, 0, 1], reused
    dev.close()


def _quad_lattice(nx, ny, seed=0):
    """nx x ny jittered lattice, horizontal + vertical edges only (degree 4: every slot row of an
    isolated tile carries one pad slot in five)."""
    rng = np.random.default_rng(seed)
    ix, iy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    pos = (np.stack([ix.ravel() * 10.0 + 5.0, iy.ravel() * 8.0 + 4.0], 1) + rng.uniform(-1, 1, (nx * ny, 2))).astype(np.float32)
    vid = (iy * nx + ix)
    eh = np.stack([vid[:, :-1].ravel(), vid[:, 1:].ravel()], 1)
    ev = np.stack([vid[:-1, :].ravel(), vid[1:, :].ravel()], 1)
    edges = np.concatenate([eh, ev]).astype(np.int32)
    d = pos[edges[:, 0]] - pos[edges[:, 1]]
    alpha = (1.0 / np.sqrt((d * d).sum(1))).astype(np.float32)
    z = (0.5 + 0.001 * pos[:, 0] + rng.normal(0, 0.02, nx * ny)).astype(np.float32)
    return pos, edges, alpha, z


def test_isolated_tile_that_does_not_fit_goes_to_the_device_builder(gpu):
    """Graphs between the isolated-tile limit and what really fits one tile.  (1) a 1 344-vertex feature
    grid with tile_single_max = 2048: the sizing rule sees that it does not fit and plans halo tiles on
    the GPU.  (2) a degree-4 lattice of 1 740 vertices passes the rule (it prices the slot rows from
    (V, E)) and fails in the builder: the answer must not be a halo plan built on the host (4 ms) --
    the graph goes to the device builder, the handle remembers the size, results are the oracle's."""
    g = graphgen.named("g15")[0]
    assert 1300 < g.V < 1400
    dev = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, tile_single_max=2048)
    assert dev.info("num_tiles") > 1 and dev.info("plan_on_device") == 1
    dev.close()

    from oracle import COracle
    pos, edges, alpha, z = _quad_lattice(30, 58)
    V = len(pos)
    wgt = np.ones(V, np.float32)
    dev = GraphRegularizer(pos, edges, alpha, alpha, z, wgt, device=0, tile_single_max=2048)
    assert dev.info("num_tiles") > 1 and dev.info("plan_on_device") == 1
    assert dev.info("single_cap") == V - 1
    o = COracle(pos, edges, alpha, alpha, z, wgt)
    o.solve(oracle_params(), 30)
    dev.step(default_params(), 30)
    x, w1, w2, q = dev.download()
    assert_bit_equal(x, o.x, "x")
    assert_bit_equal(q, o.q, "q")
    # the next graph of that size does not try the isolated tile again; a smaller one still gets it
    pos2, edges2, alpha2, z2 = _quad_lattice(30, 58, seed=1)
    dev.reupload(pos2, edges2, alpha2, alpha2, z2, wgt)
    assert dev.info("num_tiles") > 1 and dev.info("plan_on_device") == 1
    g3 = graphgen.named("g20")[0]
    dev.reupload(g3.pos, g3.edges, g3.alpha, g3.beta, g3.z, g3.wgt, tris=g3.tris)
    assert dev.info("num_tiles") == 1
    dev.close()


def test_small_frames_planned_by_one_launch(gpu):
    """Graph sync of small frames on a stream (the facade's options): from the second frame on the edges,
    the data terms and everything in front of the tile pass come from ONE launch (plan_mini).  The plan is
    the one the two dozen separate launches build -- array for array, on a second handle fed the same
    stream with plan_mini = 0 -- and the solve is the oracle's bits.  A frame that the previous partition
    does not suit (all features in one corner) falls back by itself."""
    from flame_ros_amd.regularizer import default_sync_params
    from oracle import COracle
    sp, p = default_sync_params(), default_params()
    names = ("v_o2i", "v_i2o", "e_o2i", "e_i2o", "grow", "ginc", "tris", "trow", "tinc", "t_vmap", "t_emap", "t_srow")
    a = GraphRegularizer.empty(device=0, tile_single_max=1, stream_depth=5)
    b = GraphRegularizer.empty(device=0, tile_single_max=1, stream_depth=5, plan_mini=0)
    rng = np.random.default_rng(11)
    used = []
    for k in range(9):
        g = graphgen.named("tum" if k < 6 else "v2000", seed=20 + k)[0]
        if k == 4:  # scene change: every feature inside the top-left eighth of the image
            pos = (rng.random((1200, 2)) * np.array([80.0, 60.0])).astype(np.float32)
            g = graphgen.from_points(pos, 640, 480, np.random.Generator(np.random.PCG64(5)))
        var = np.full(g.V, 1e-4, np.float32)
        pred = (g.z * 1.01).astype(np.float32) if k & 1 else None
        for r in (a, b):
            r.sync_features(g.pos, g.z, var, g.tris, sp, prediction=pred)
        used.append((a.info("plan_mini"), a.info("plan_reused")))
        assert b.info("plan_mini") == 0 and a.info("plan_on_device") == 1 and b.info("plan_on_device") == 1
        assert a.info("plan_reused") == b.info("plan_reused"), k
        for nm in names:
            assert np.array_equal(a.plan_array(nm, np.int32), b.plan_array(nm, np.int32)), (k, nm)
        for nm in ("eij", "ew", "t_eij", "t_ew"):
            assert np.array_equal(a.plan_array(nm, np.uint32), b.plan_array(nm, np.uint32)), (k, nm)
        e = a.edges()
        assert np.array_equal(e, g.edges), k
        o = COracle(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, x0=pred)
        o.solve(oracle_params(), 25)
        a.step(p, 25)
        x, w1, w2, q = a.download()
        assert_bit_equal(x, o.x, "frame %d x" % k)
        assert_bit_equal(q, o.q, "frame %d q" % k)
    # (plan_mini, plan_reused): first frame; three from the one launch; the corner frame -- edges and data
    # terms from the one launch, its partition rejected and bisected; one frame of back-off; a new size;
    # and the one launch again
    assert used == [(0, 0), (1, 1), (1, 1), (1, 1), (1, 0), (0, 0), (0, 0), (1, 1), (1, 1)], used
    a.close(); b.close()


def test_small_odd_meshes_through_one_launch(gpu):
    """Small frames that are not triangulated disks, every frame three times in a row (the third time its edge
    count is predicted from the offset seen before, which is what admits it to plan_mini): a hub of degree
    72 (rows longer than the register sorts), holes, vertices no triangle references, a triangle listed
    twice, two components.  Edges, plan arrays (against a handle with plan_mini = 0) and the solve
    (against the oracle's sync + iterations) are the same bits."""
    from flame_ros_amd.regularizer import default_sync_params
    from oracle import COracle
    from oracle.cbind import SyncParams as OSync, graph_sync as oracle_sync
    sp, p = default_sync_params(), default_params()
    names = ("v_o2i", "v_i2o", "e_o2i", "e_i2o", "grow", "ginc", "tris", "trow", "tinc", "t_vmap", "t_emap", "t_srow")
    a = GraphRegularizer.empty(device=0, tile_single_max=1, stream_depth=5)
    b = GraphRegularizer.empty(device=0, tile_single_max=1, stream_depth=5, plan_mini=0)
    rng = np.random.default_rng(23)
    took = {}
    order = ["disk", "hub", "holes", "isolated", "twice", "two_parts"]
    # the same frame three times: a wrong edge-count guess, one frame of back-off, then the offset predicts it
    kinds = ["disk"] + [kk for kk in order[1:] for _ in range(3)] + ["disk"]
    for k, kind in enumerate(kinds):
        rng = np.random.default_rng(23 + order.index(kind))
        pos = (rng.random((1300, 2)) * np.array([640.0, 480.0])).astype(np.float32)
        if kind == "hub":  # 72 points on a circle, nothing but the centre inside
            c = np.array([320.0, 240.0])
            keep = np.linalg.norm(pos - c, axis=1) > 81.0
            ang = np.arange(72) * (2 * np.pi / 72)
            ring = c + 80.0 * np.column_stack([np.cos(ang), np.sin(ang)])
            pos = np.concatenate([pos[keep], ring, c[None]]).astype(np.float32)
        g = graphgen.from_points(pos, 640, 480, np.random.Generator(np.random.PCG64(40 + order.index(kind))))
        tris = g.tris
        if kind == "hub":
            deg = np.bincount(g.edges.ravel(), minlength=g.V)
            assert deg[-1] == 72
        elif kind == "holes":
            tris = tris[rng.random(len(tris)) > 0.15]
        elif kind == "isolated":  # every triangle of 40 vertices goes: they stay in the frame without edges
            gone = rng.choice(g.V, 40, replace=False)
            tris = tris[~np.isin(tris, gone).any(1)]
        elif kind == "twice":
            tris = np.concatenate([tris, tris[100:103]])
        elif kind == "two_parts":
            cx = g.pos[tris].mean(1)[:, 0]
            tris = tris[(cx < 300) | (cx > 330)]
        tris = np.ascontiguousarray(tris, np.int32)
        var = np.full(g.V, 1e-4, np.float32)
        s = oracle_sync(OSync(0, 0, 1, 0.01), g.pos, g.z, var, tris, None)
        for r in (a, b):
            r.sync_features(g.pos, g.z, var, tris, sp)
        took.setdefault(kind, []).append(a.info("plan_mini"))
        assert b.info("plan_mini") == 0 and a.E == len(s["edges"]) == b.E, (k, kind)
        assert np.array_equal(a.edges(), s["edges"]), (k, kind)
        if a.info("plan_reused") == b.info("plan_reused"):
            for nm in names:
                assert np.array_equal(a.plan_array(nm, np.int32), b.plan_array(nm, np.int32)), (k, kind, nm)
        o = COracle(g.pos, s["edges"], s["alpha"], s["beta"], s["z"], s["wgt"], x0=s["x0"])
        o.solve(oracle_params(), 20)
        for r in (a, b):
            r.step(p, 20)
            x, w1, w2, q = r.download()
            assert_bit_equal(x, o.x, "frame %d (%s) x" % (k, kind))
            assert_bit_equal(q, o.q, "frame %d (%s) q" % (k, kind))
    print("plan_mini per kind:", took)
    assert all(v[-1] == 1 for kk, v in took.items() if kk != "disk"), took
    a.close(); b.close()


def test_stream_depth_option(gpu):
    """stream_depth replaces the automatic depth 8 of small graphs (<= 2048 vertices) solved by launches (persist = 0;
    resident tiles choose depth 5 there by themselves) and nothing else."""
    p = default_params()
    for name, want in (("v2000", 5), ("5k", None)):
        g = graphgen.named(name)[0]
        a = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, persist=0)
        b = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, stream_depth=5, persist=0)
        assert b.info("tile_depth") == (want if want else a.info("tile_depth"))
        if want:
            assert a.info("tile_depth") == 8
        a.step(p, 23); b.step(p, 23)
        assert_bit_equal(a.download()[0], b.download()[0], name)
        a.close(); b.close()


def test_device_plan_growing_frames(gpu):
    """Frames that grow threefold each on ONE handle: every scratch buffer of the builder (lists,
    counters, page-locked landing areas, tile arrays) is re-reserved on the way."""
    host = dev = None
    for k, V in enumerate((3000, 9000, 27000, 2500)):
        g = graphgen.synthetic(V, seed=60 + k)
        if host is None:
            host = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=-1, **host_reference_opts())
            dev = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, plan_reuse=0)
        else:
            host.reupload(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris)
            dev.reupload(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris)
        compare_plans(host, dev, "growing frame %d (V=%d)" % (k, V))
        o = make_oracle(g)
        o.solve(oracle_params(), 17)
        dev.step(default_params(), 17)
        assert_bit_equal(dev.download()[0], o.x, "growing frame %d x" % k)
    host.close(); dev.close()


def test_device_plan_subtree_overflow_recovery(gpu):
    """The LDS subtree kernel of the bisection reports an overflow (forced here through the
    debug_sub_cap test hook; in production: very uneven weighted splits): the builder hands over
    one level later and the plan still equals the host builder's."""
    g, _ = graphgen.named("50k")
    host = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=-1, **host_reference_opts())
    dev = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, debug_sub_cap=2000, plan_reuse=0)
    compare_plans(host, dev, "overflow recovery")
    host.reupload(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris)
    dev.reupload(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris)
    compare_plans(host, dev, "after recovery")
    host.close(); dev.close()


def test_device_plan_error_conventions(gpu):
    """Bad indices and non-finite inputs are found by the device builder's own checks."""
    g = graphgen.synthetic(4000, seed=9)
    bad = g.edges.copy(); bad[17, 1] = g.V
    with pytest.raises(_l.FlameHipError) as e:
        GraphRegularizer(g.pos, bad, g.alpha, g.beta, g.z, g.wgt, tris=g.tris)
    assert e.value.code == _l.ERR_ARG
    bad = g.tris.copy(); bad[5, 2] = -1
    with pytest.raises(_l.FlameHipError) as e:
        GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=bad)
    assert e.value.code == _l.ERR_ARG
    z = g.z.copy(); z[100] = np.inf
    with pytest.raises(_l.FlameHipError) as e:
        GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, z, g.wgt, tris=g.tris)
    assert e.value.code == _l.ERR_NAN
    a = g.alpha.copy(); a[3] = np.nan
    with pytest.raises(_l.FlameHipError) as e:
        GraphRegularizer(g.pos, g.edges, a, g.beta, g.z, g.wgt, tris=g.tris)
    assert e.value.code == _l.ERR_NAN


def test_host_plan_still_selectable(gpu):
    g = graphgen.synthetic(6000, seed=3)
    o = make_oracle(g)
    o.solve(oracle_params(), 40)
    with GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, plan_device=0) as r:
        assert r.info("plan_on_device") == 0 and r.info("path") == 2
        r.step(default_params(), 40)
        assert_bit_equal(r.download(with_q=False)[0], o.x, "host plan x")
