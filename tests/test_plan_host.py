"""CPU: the C ABI surface and the host-side plan (locality order, incidence CSR, tile partition with
depth-D halos).  No GPU: handles are created with device = -1 ("plan only"); every compute entry
point must then refuse with FLAME_HIP_ERR_NODEVICE -- the product has no CPU fallback."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from flame_ros_amd import lib
from flame_ros_amd.regularizer import GraphRegularizer, default_params, default_tri_params
from oracle.nltgv2_np import NpSolver
from tests.util import graphgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "flame_hip.h")).read()
    declared = set(re.findall(r"\b(flame_hip_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(lib.SYMBOLS), declared ^ set(lib.SYMBOLS)
    L = lib.load()
    for name in declared:
        assert getattr(L, name) is not None
    assert L.flame_hip_version() >= 100
    assert L.flame_hip_strerror(0) == b"ok" and b"device" in L.flame_hip_strerror(lib.ERR_NODEVICE)


def test_struct_layouts_match_the_header():
    assert C.sizeof(lib.Params) == 24 and C.sizeof(lib.TriParams) == 40
    assert C.sizeof(lib.TileDesc) == 4 * (13 + 17 + 17)


def test_no_cpu_fallback():
    g = graphgen.synthetic(300, seed=1)
    r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=-1)
    p = default_params()
    import ctypes as C2
    L = lib.load()
    gave = C2.c_int32()
    assert L.flame_hip_state_snapshot(r._h, None) == lib.ERR_NODEVICE and L.flame_hip_state_rollback(r._h, None) == lib.ERR_NODEVICE
    assert L.flame_hip_persist_take_error(r._h, C2.byref(gave)) == lib.ERR_NODEVICE
    tcount = C2.c_int32()
    assert L.flame_hip_delaunay(r._h, 3, g.pos.ctypes.data_as(C2.c_void_p), 0, None, C2.byref(tcount)) == lib.ERR_STATE  # (plan-only handle: no GPU)
    assert L.flame_hip_delaunay_list(r._h, 0, None) == lib.ERR_STATE
    for call in (lambda: r.step(p, 1), lambda: r.costs(p), lambda: r.download(),
                 lambda: r.set_state(x=g.z), lambda: r.sync(),
                 lambda: r.triangles(np.eye(3), default_tri_params())):
        with pytest.raises(lib.FlameHipError) as e:
            call()
        assert e.value.code == lib.ERR_NODEVICE


def test_argument_errors():
    g = graphgen.synthetic(100, seed=2)
    with pytest.raises(lib.FlameHipError) as e:
        bad = g.edges.copy(); bad[3, 1] = 100
        GraphRegularizer(g.pos, bad, g.alpha, g.beta, g.z, g.wgt, device=-1)
    assert e.value.code == lib.ERR_ARG
    with pytest.raises(lib.FlameHipError) as e:
        bad = g.edges.copy(); bad[3, 1] = bad[3, 0]
        GraphRegularizer(g.pos, bad, g.alpha, g.beta, g.z, g.wgt, device=-1)
    assert e.value.code == lib.ERR_ARG
    with pytest.raises(lib.FlameHipError) as e:
        z = g.z.copy(); z[5] = np.nan
        GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, z, g.wgt, device=-1)
    assert e.value.code == lib.ERR_NAN
    with pytest.raises(lib.FlameHipError) as e:
        GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, device=-1, no_such_option=1)
    assert e.value.code == lib.ERR_ARG
    with pytest.raises(lib.FlameHipError) as e:
        GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, device=-1, tile_threads=100)
    assert e.value.code == lib.ERR_ARG
    with pytest.raises(ValueError):
        GraphRegularizer(g.pos, g.edges, g.alpha[:-1], g.beta, g.z, g.wgt, device=-1)
    L = lib.load()
    assert L.flame_hip_graph_create(None, 0, 1, 1, 0) == lib.ERR_ARG
    h = C.c_void_p()
    assert L.flame_hip_graph_create(C.byref(h), -1, -5, 0, 0) == lib.ERR_ARG
    assert L.flame_hip_graph_create(C.byref(h), -1, 10, 0, 0) == 0
    assert L.flame_hip_solve(h, C.byref(default_params()), 1, None) == lib.ERR_STATE  # no upload
    L.flame_hip_graph_destroy(h)
    L.flame_hip_graph_destroy(None)


def tiles_of(r):
    raw = r.plan_array("tiles", np.dtype((np.void, C.sizeof(lib.TileDesc))))
    return (lib.TileDesc * len(raw)).from_buffer_copy(raw.tobytes())


PLANS = [(2000, dict(tile_own=64, tile_depth=3)), (3000, dict(tile_own=200, tile_depth=4)),
         (1500, {}), (700, dict(tile_own=50, tile_depth=1)), (5000, dict(tile_own=96, tile_depth=5))]


@pytest.mark.parametrize("V,opts", PLANS)
def test_plan_invariants(V, opts):
    g = graphgen.synthetic(V, seed=V)
    # the sorted local edge order (lane_order = 0); the conflict-avoiding lane order is checked
    # against it in test_lane_order_permutes_inside_blocks
    r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=-1, lane_order=0, **opts)
    assert r.info("path") == lib.PATH_TILE
    v_o2i, v_i2o = r.plan_array("v_o2i", np.int32), r.plan_array("v_i2o", np.int32)
    e_o2i, e_i2o = r.plan_array("e_o2i", np.int32), r.plan_array("e_i2o", np.int32)
    assert np.array_equal(np.sort(v_o2i), np.arange(g.V)) and np.array_equal(v_i2o[v_o2i], np.arange(g.V))
    assert np.array_equal(np.sort(e_o2i), np.arange(g.E)) and np.array_equal(e_i2o[e_o2i], np.arange(g.E))
    eij = r.plan_array("eij", np.int32).reshape(-1, 2)
    assert np.array_equal(eij, v_o2i[g.edges[e_i2o]])  # orientation preserved
    grow, ginc = r.plan_array("grow", np.int32), r.plan_array("ginc", np.int32)
    assert grow[0] == 0 and grow[-1] == 2 * g.E
    for v in range(0, g.V, 97):
        ent = ginc[grow[v]:grow[v + 1]]
        k = ent & 0x7fffffff
        orig = e_i2o[k]
        assert np.all(np.diff(orig) > 0)  # ascending ORIGINAL edge id: the oracle's scatter order
        assert np.all(np.where(ent < 0, eij[k, 1], eij[k, 0]) == v)

    tiles = tiles_of(r)
    depth = r.info("tile_depth")
    vmap, emap = r.plan_array("t_vmap", np.int32), r.plan_array("t_emap", np.int32)
    t_eij = r.plan_array("t_eij", np.uint32).reshape(-1, 2)
    srow = r.plan_array("t_srow", np.uint32)
    own_cover = np.zeros(g.V, np.int32)
    edge_cover = np.zeros(g.E, np.int32)
    deg = np.diff(grow)
    import scipy.sparse as sp
    from scipy.sparse.csgraph import dijkstra
    A = sp.coo_matrix((np.ones(g.E), (eij[:, 0], eij[:, 1])), shape=(g.V, g.V))
    A = (A + A.T).tocsr()
    for ti, T in enumerate(tiles):
        own_cover[T.vstart:T.vstart + T.n_own] += 1
        edge_cover[T.estart:T.estart + T.e_own] += 1
        ext = vmap[T.vmap_off:T.vmap_off + T.n_ext]
        assert np.array_equal(ext[:T.n_own], np.arange(T.vstart, T.vstart + T.n_own))
        assert len(set(ext.tolist())) == T.n_ext
        loc_e = emap[T.emap_off:T.emap_off + T.e_loc]
        assert np.array_equal(loc_e[:T.e_own], np.arange(T.estart, T.estart + T.e_own))
        assert np.all(np.isin(eij[T.estart:T.estart + T.e_own, 0], ext[:T.n_own]))  # owner = source's tile
        rec = t_eij[T.erec_off:T.erec_off + T.e_loc]
        li, lj = rec[:, 0] & 0xffff, rec[:, 0] >> 16
        assert np.array_equal(ext[li], eij[loc_e, 0]) and np.array_equal(ext[lj], eij[loc_e, 1])
        ring_end, level_end = list(T.ring_end), list(T.level_end)
        assert ring_end[0] == T.n_own and ring_end[depth] == T.n_ext and level_end[depth] == T.e_loc
        ring = np.searchsorted(np.array(ring_end[:depth + 1]), np.arange(T.n_ext), side="right")
        if ti % 7 == 0:  # rings are exact graph distances from the own set
            dist = dijkstra(A, unweighted=True, indices=ext[:T.n_own], min_only=True, limit=depth + 1)
            assert np.array_equal(dist[ext].astype(int), ring)
            # halo closure: every neighbour of a vertex at ring < depth is in the tile
            inner = ext[ring < max(depth, 1)] if depth else ext
            nb = A[inner].indices
            assert np.all(np.isin(nb, ext))
        level = np.maximum(ring[li], ring[lj])
        assert np.all(np.diff(level) >= 0)
        own_e = slice(0, T.e_own)  # owned edges: grouped by source (adjacent lanes share gathers)
        assert np.all(np.diff(li[own_e].astype(np.int64))[level[own_e][1:] == level[own_e][:-1]] >= 0)
        for l in range(depth + 1):
            assert level_end[l] == int((level <= l).sum())
        assert T.n_upd == (T.n_ext if depth == 0 else ring_end[depth - 1])
        # incidence slots: every (updated vertex, incidence) is hit by exactly one edge endpoint,
        # at position = rank of the edge's original id in the vertex's list (row-major, odd pitch)
        sr = srow[T.srow_off:T.srow_off + T.n_upd]
        s0, dg = (sr & 0xffff).astype(np.int64), (sr >> 16).astype(np.int64)
        assert np.array_equal(dg, deg[ext[:T.n_upd]])
        ss, sd = (rec[:, 1] & 0xffff).astype(np.int64), (rec[:, 1] >> 16).astype(np.int64)
        used = {}
        for e in range(T.e_loc):
            for lv, slot in ((li[e], ss[e]), (lj[e], sd[e])):
                if lv < T.n_upd:
                    assert slot != 0xffff and 0 <= slot - s0[lv] < dg[lv] and slot < T.nslots
                    used.setdefault(int(lv), []).append((slot - s0[lv], e_i2o[loc_e[e]]))
                else:
                    assert slot == 0xffff
        for lv, lst in used.items():
            lst.sort()
            assert [p for p, _ in lst] == list(range(dg[lv]))
            assert all(a[1] < b[1] for a, b in zip(lst, lst[1:]))
        assert len(used) == int((dg > 0).sum())
    assert np.all(own_cover == 1) and np.all(edge_cover == 1)
    assert r.info("tile_lds_bytes") <= r.info("lds_bytes")
    assert r.info("tile_threads") * r.info("tile_ept") >= max(t.e_loc for t in tiles)
    assert r.info("tile_threads") * r.info("tile_vpt") >= max(t.n_ext for t in tiles)


def _read_groups():
    g = []
    for lane in range(64):
        h = lane & 31
        g.append((0 if (h < 4 or 12 <= h < 16 or 20 <= h < 28) else 1) + 2 * (lane >> 5))
    return np.array(g)


def _phase_d_conflicts(rec, nslots):
    """Modelled extra LDS cycles of one 64-edge block (MI355X guide, LDS): ds_read_b128 in 4 groups
    of 16 lanes with 16 bank classes, ds_write_b96 in 8 groups of 8 lanes with 8 classes; equal
    addresses broadcast; a group costs its busiest class."""
    c = len(rec)
    lanes = np.arange(c)
    li, lj = rec[:, 0] & 0xffff, rec[:, 0] >> 16
    ss, sd = rec[:, 1] & 0xffff, rec[:, 1] >> 16
    ss = np.where(ss == 0xffff, nslots + lanes, ss)
    sd = np.where(sd == 0xffff, nslots + lanes, sd)
    rg = _read_groups()[:c]
    extra = 0
    for idx, grp, mod in ((li, rg, 16), (lj, rg, 16), (ss, lanes >> 3, 8), (sd, lanes >> 3, 8)):
        for q in np.unique(grp):
            a = np.unique(idx[grp == q])
            extra += int(np.bincount(a % mod, minlength=mod).max()) - 1
    return extra


@pytest.mark.parametrize("V,opts", PLANS[:3])
def test_lane_order_permutes_inside_blocks(V, opts):
    """lane_order = 2 (at build time; the default 1 applies the same order on the device when a plan
    is solved a second time): inside every block of 64 local edges (one wave's share of phase D)
    the edges are re-assigned to lanes against LDS bank conflicts; which edges a block holds, and
    every other plan array, is unchanged, and the modelled conflict cycles go down."""
    g = graphgen.synthetic(V, seed=V)
    r0 = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=-1, lane_order=0, **opts)
    r1 = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=-1, lane_order=2, **opts)
    rd = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=-1, **opts)
    for name, dt in (("t_emap", np.int32), ("t_eij", np.uint32)):  # the default leaves the sorted order at build
        assert np.array_equal(r0.plan_array(name, dt), rd.plan_array(name, dt)), name
    for name, dt in (("v_o2i", np.int32), ("e_o2i", np.int32), ("grow", np.int32), ("ginc", np.int32),
                     ("tiles", np.int32), ("t_vmap", np.int32), ("t_srow", np.uint32)):
        assert np.array_equal(r0.plan_array(name, dt), r1.plan_array(name, dt)), name
    e0, e1 = r0.plan_array("t_emap", np.int32), r1.plan_array("t_emap", np.int32)
    c0 = np.concatenate([r0.plan_array("t_eij", np.uint32).reshape(-1, 2), r0.plan_array("t_ew", np.uint32).reshape(-1, 4)], axis=1)
    c1 = np.concatenate([r1.plan_array("t_eij", np.uint32).reshape(-1, 2), r1.plan_array("t_ew", np.uint32).reshape(-1, 4)], axis=1)
    before = after = 0
    for T in tiles_of(r0):
        for b0 in range(0, T.e_loc, 64):
            s = slice(T.emap_off + b0, T.emap_off + min(b0 + 64, T.e_loc))
            p = np.argsort(e1[s], kind="stable")
            q = np.argsort(e0[s], kind="stable")
            assert np.array_equal(e1[s][p], e0[s][q])            # the same edges ...
            assert np.array_equal(c1[s][p], c0[s][q])            # ... with their records
            before += _phase_d_conflicts(c0[s, :2], T.nslots)
            after += _phase_d_conflicts(c1[s, :2], T.nslots)
    assert after < 0.8 * before, (before, after)


def test_tile_schedule_reproduces_global_iteration():
    """Emulates what a tile launch does (d unmasked PD iterations on the tile's local subgraph,
    float64) and checks that what the tile OWNS equals the global iteration after d steps: pins
    the halo-depth logic (rings, local edge set, ownership) without a GPU."""
    g = graphgen.dataset_shaped(320, 240, 8, seed=3)
    d = 3
    r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, device=-1, tile_own=60, tile_depth=d)
    v_i2o, e_i2o = r.plan_array("v_i2o", np.int32), r.plan_array("e_i2o", np.int32)
    vmap, emap = r.plan_array("t_vmap", np.int32), r.plan_array("t_emap", np.int32)
    ref = NpSolver(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt)
    ref.solve(7)  # generic state
    st = dict(x=ref.x.copy(), w=ref.w.copy(), xb=ref.xb.copy(), wb=ref.wb.copy(), q=ref.q.copy())
    ref.solve(d)
    for T in list(tiles_of(r))[::3]:
        ext_o = v_i2o[vmap[T.vmap_off:T.vmap_off + T.n_ext]]
        loc_o = e_i2o[emap[T.emap_off:T.emap_off + T.e_loc]]
        lid = -np.ones(g.V, np.int64)
        lid[ext_o] = np.arange(T.n_ext)
        s = NpSolver(g.pos[ext_o], lid[g.edges[loc_o]], g.alpha[loc_o], g.beta[loc_o], g.z[ext_o], g.wgt[ext_o])
        s.x, s.w, s.xb, s.wb, s.q = st["x"][ext_o], st["w"][ext_o], st["xb"][ext_o], st["wb"][ext_o], st["q"][loc_o]
        s.solve(d)
        own = slice(0, T.n_own)
        assert np.abs(s.x[own] - ref.x[ext_o[own]]).max() < 1e-13
        assert np.abs(s.xb[own] - ref.xb[ext_o[own]]).max() < 1e-13
        assert np.abs(s.w[own] - ref.w[ext_o[own]]).max() < 1e-13
        assert np.abs(s.q[:T.e_own] - ref.q[loc_o[:T.e_own]]).max() < 1e-12
        if T.n_ext > T.n_own:  # and the outer ring really is contaminated (the halo is needed)
            assert np.abs(s.x - ref.x[ext_o]).max() > 1e-9


def test_global_path_and_fallbacks():
    g = graphgen.synthetic(3000, seed=5)
    r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, device=-1, path=1)
    assert r.info("path") == lib.PATH_GLOBAL and r.info("num_tiles") == 0
    # a graph that fits one LDS tile is a single isolated tile (depth 0: any iterations/launch)
    g = graphgen.dataset_shaped(320, 240, 16)  # 300 vertices: small enough to be one tile by default
    r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, device=-1)
    assert r.info("num_tiles") == 1 and r.info("tile_depth") == 0
    g = graphgen.dataset_shaped(640, 480, 16)  # 1200 vertices: tiles by default, one tile on request
    r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, device=-1)
    assert r.info("num_tiles") > 1 and r.info("tile_depth") == 5   # (resident tiles, the default: shallow halos)
    r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, device=-1, persist=0)
    assert r.info("num_tiles") > 1 and r.info("tile_depth") == 8   # (launches: deep halos amortise the kernel boundary)
    r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, device=-1, tile_own=g.V)
    assert r.info("num_tiles") == 1 and r.info("tile_depth") == 0
    # empty graph
    r = GraphRegularizer(np.zeros((0, 2)), np.zeros((0, 2), np.int32), [], [], [], [], device=-1)
    assert r.info("V") == 0


def test_isolated_tile_sizing_prices_the_slot_rows():
    """One isolated tile holds 16 B per vertex + 16 B per incidence SLOT; the slot rows of a 64-vertex group
    share the group's largest degree (odd) as pitch.  The sizing rule prices that from (V, E): a 1 344-
    vertex feature grid is not offered as one tile any more (it used to be, failed in the builder and was
    rebuilt as a halo plan on the host: 4.6 ms per frame), 1 200 vertices still are one tile on request;
    a degree-4 lattice passes the rule and fails in the builder (one pad slot in five): the plan-only
    builder falls back to halo tiles by itself, with a valid plan."""
    g = graphgen.named("g15")[0]  # 1 344 vertices
    r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, device=-1, tile_single_max=2048)
    assert r.info("num_tiles") > 1
    g = graphgen.named("tum")[0]  # 1 200 vertices
    r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, device=-1, tile_single_max=2048)
    assert r.info("num_tiles") == 1 and r.info("tile_depth") == 0
    # 30 x 58 lattice, horizontal + vertical edges
    nx, ny = 30, 58
    ix, iy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    pos = np.stack([ix.ravel() * 10.0 + 5.0, iy.ravel() * 8.0 + 4.0], 1).astype(np.float32)
    vid = iy * nx + ix
    edges = np.concatenate([np.stack([vid[:, :-1].ravel(), vid[:, 1:].ravel()], 1),
                            np.stack([vid[:-1, :].ravel(), vid[1:, :].ravel()], 1)]).astype(np.int32)
    V, E = len(pos), len(edges)
    assert V * 16 + E * 38 + 1024 <= 160 * 1024  # the rule says "fits"
    alpha = np.ones(E, np.float32)
    r = GraphRegularizer(pos, edges, alpha, alpha, np.ones(V, np.float32), np.ones(V, np.float32), device=-1,
                         tile_single_max=2048)
    assert r.info("path") == lib.PATH_TILE and r.info("num_tiles") > 1
    v_o2i = r.plan_array("v_o2i", np.int32)
    assert np.array_equal(np.sort(v_o2i), np.arange(V))


def test_fat_tile_sizing():
    """r05: beyond 256 x 196 vertices a handle that solves by resident tiles keeps ONE tile per CU: the halo gets shallower as
    the tiles grow, a tile may hold more local vertices than threads (only updated vertices and poll slots need a lane), and
    the incidence slots shrink to 12 bytes where 16 do not fit 160 KiB (the resident staging area included).  Without resident
    tiles, or with a forced tile size, the r04 partition (two rounds of smaller tiles) stays."""
    for V, depth, s12 in ((60000, 4, 0), (100000, 3, 0), (160000, 1, 0), (200000, 2, 1)):
        g = graphgen.synthetic(V, seed=V)
        r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, device=-1)
        tiles = tiles_of(r)
        assert len(tiles) == 256 and r.info("tile_depth") == depth and r.info("tile_slot12") == s12, (V, len(tiles), r.info("tile_depth"))
        nt, ept, vpt = r.info("tile_threads"), r.info("tile_ept"), r.info("tile_vpt")
        assert vpt == 1 and ept in (2, 3) and (nt == 1024 or not s12)  # a configuration the resident kernels have
        assert max(t.e_loc for t in tiles) <= nt * ept
        assert max(t.n_upd for t in tiles) <= nt and max(t.n_ext - t.n_own for t in tiles) <= nt
        stage = max(16 * max(t.n_upd - t.n_own, 0) for t in tiles)
        per_slot = 12 if s12 else 16
        need = max(16 * t.n_ext + per_slot * ((t.nslots + 64 + 1 + (3 if s12 else 0)) // (4 if s12 else 1) * (4 if s12 else 1)) for t in tiles)
        assert need == r.info("tile_lds_bytes") and need + stage + 4096 <= 160 * 1024  # (fat tiles keep a margin)
        if V == 200000:
            assert max(t.n_ext for t in tiles) > nt  # the lane-less outermost ring
        if V == 100000:
            r0 = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, device=-1, persist=0)
            assert r0.info("num_tiles") > 256 and r0.info("tile_slot12") == 0 and r0.info("tile_depth") == 3
            r1 = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, device=-1, tile_own=300)
            assert r1.info("num_tiles") > 256 and r1.info("tile_slot12") == 0


def test_one_xcd_sizing():
    """r06: a resident graph of 770 .. 1 280 vertices takes 32 tiles (an XCD has 32 CUs: the tiles hand over through its L2,
    option "one_xcd"); below, the 24-vertex floor already gives at most 32; above, and with the option off or without resident
    tiles, the sizes of all XCDs apply."""
    for V, ntiles in ((700, 30), (800, 32), (1200, 32), (1280, 32), (1300, 55), (1500, 63)):
        g = graphgen.synthetic(V, seed=V)
        r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, device=-1)
        assert r.info("num_tiles") == ntiles, (V, r.info("num_tiles"))
        assert r.info("tile_threads") == 512 and r.info("tile_ept") in (2, 3) and r.info("tile_vpt") == 1  # a resident configuration
    g = graphgen.synthetic(1200, seed=1200)
    assert GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, device=-1, one_xcd=0).info("num_tiles") == 50
    assert GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, device=-1, persist=0).info("num_tiles") == 38
    assert GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, device=-1, tile_own=30).info("num_tiles") == 40


def test_peer_transport_inboxes_are_consistent_across_ranks():
    """r06 peer transport: every rank derives every rank's inbox layout from the whole graph (no request lists travel).  Host-only
    plans of the 3 ranks of a 3 x 2 partition: what rank a sends to part d is exactly what the rank of d expects from it."""
    from flame_ros_amd import partition
    g = graphgen.synthetic(6000, seed=3)
    world, k = 3, 2
    plans = [partition.Partition(None, g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, parts_per_rank=k, halo_depth=4, plan_rank=r, plan_world=world)
             for r in range(world)]
    info = {}
    for r, ps in enumerate(plans):
        for i in range(k):
            me = ps.info("part_id", i)
            peers, sc, rc = ps.array("peers", i), ps.array("send_cnt", i).reshape(-1, 2), ps.array("recv_cnt", i).reshape(-1, 2)
            info[me] = {int(p): (tuple(sc[j]), tuple(rc[j])) for j, p in enumerate(peers)}
    assert sorted(info) == list(range(world * k))
    for a, d in info.items():
        for b, (sent, recv) in d.items():
            assert a in info[b], (a, b)          # the peer relation is symmetric
            assert info[b][a][1] == sent and info[b][a][0] == recv, (a, b, sent, recv, info[b][a])


def test_batch_plan():
    gs = [graphgen.dataset_shaped(640, 480, 16, seed=s) for s in range(3)] + [graphgen.synthetic(200, seed=1)]
    r = GraphRegularizer.from_batch(gs, device=-1)
    tiles = tiles_of(r)
    assert len(tiles) == 4 and r.info("tile_depth") == 0
    for b, T in enumerate(tiles):
        assert (T.vstart, T.n_own, T.n_ext, T.e_own, T.e_loc) == (r.voff[b], gs[b].V, gs[b].V, gs[b].E, gs[b].E)
    # an edge crossing two frames, and a frame too large for one tile, are argument errors
    g0, g1 = gs[0], gs[3]
    voff = np.array([0, g0.V, g0.V + g1.V], np.int32)
    edges = np.concatenate([g0.edges, g1.edges + g0.V]).astype(np.int32)
    edges[0, 1] = g0.V + 1
    cat = lambda n: np.concatenate([getattr(g0, n), getattr(g1, n)])  # noqa: E731
    with pytest.raises(lib.FlameHipError) as e:
        GraphRegularizer(cat("pos"), edges, cat("alpha"), cat("beta"), cat("z"), cat("wgt"), device=-1,
                         _batch_voff=voff)
    assert e.value.code == lib.ERR_ARG
    with pytest.raises(lib.FlameHipError) as e:
        GraphRegularizer.from_batch([graphgen.synthetic(5000, seed=2)], device=-1)
    assert e.value.code == lib.ERR_ARG
