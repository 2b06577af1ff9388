"""Host-side mirror of upstream's regulariser interface over the C ABI.

Names follow upstream `flame::optimizers::nltgv2_l1_graph_regularizer` ([UPSTREAM-RECALL], the
parameters are pinned by reference src/flame_offline_tum.cc:242-245): `Params(data_factor, step_x,
step_q, theta)`, `step()`, `smoothnessCost()`, `dataCost()`.  Everything computes on the GPU
through libflame_hip.so; there is no CPU path in this module.
"""
import ctypes as C

import numpy as np

from . import lib as _l
from .lib import Params, SyncParams, TriParams, FlameHipError  # noqa: F401


def default_params(data_factor=0.15, step_x=1e-3, step_q=125.0, theta=0.25, x_min=0.0, x_max=10.0):
    """Defaults of cfg/flame_offline_tum.yaml:93-96 (reference)."""
    return Params(data_factor, step_x, step_q, theta, x_min, x_max)


def default_tri_params(width=640, height=480):
    """Defaults of cfg/flame_offline_tum.yaml:38-53 (reference)."""
    return TriParams(1, 1.57, 0.35, 0.1, 1, 0.333, 1, 0.01, width, height)


def default_sync_params(adaptive_data_weights=False, rescale_data=False, init_with_prediction=True,
                        idepth_var_max_graph=0.01, edge_weight_rule=0, alpha_gain=0.0, beta_gain=0.0):
    """Defaults of cfg/flame_offline_tum.yaml:89-92 (reference); the last three are the
    [UPSTREAM-RECALL] switches of include/flame_hip.h (0 = the build's default statement)."""
    return SyncParams(int(adaptive_data_weights), int(rescale_data), int(init_with_prediction),
                      idepth_var_max_graph, int(edge_weight_rule), float(alpha_gain), float(beta_gain))


def feature_gate(idepth_var, var_max):
    """Row a7 gate: which tracked features may enter the graph (var < idepth_var_max_graph)."""
    var = _f32(idepth_var)
    keep = np.empty(len(var), np.uint8)
    n = _l.load().flame_hip_feature_gate(len(var), _ptr(var), var_max, _ptr(keep))
    if n < 0:
        raise FlameHipError(int(n), "flame_hip_feature_gate")
    return keep.astype(bool)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class GraphRegularizer:
    """One Delaunay vertex graph resident on one GPU (a `flame_hip_graph` handle)."""

    @classmethod
    def from_batch(cls, graphs, device=0, **options):
        """Frames axis: independent graphs (objects with pos/edges/alpha/beta/z/wgt/tris) in ONE
        handle, one LDS tile (= one workgroup) per graph, all iterations in one launch."""
        voff = np.zeros(len(graphs) + 1, np.int32)
        voff[1:] = np.cumsum([len(g.z) for g in graphs])
        toff = np.cumsum([0] + [len(g.tris) for g in graphs])
        cat = lambda name: np.concatenate([np.asarray(getattr(g, name)) for g in graphs])  # noqa: E731
        edges = np.concatenate([np.asarray(g.edges, np.int32) + voff[b] for b, g in enumerate(graphs)])
        tris = np.concatenate([np.asarray(g.tris, np.int32) + voff[b] for b, g in enumerate(graphs)])
        self = cls(cat("pos"), edges, cat("alpha"), cat("beta"), cat("z"), cat("wgt"), tris=tris,
                   device=device, _batch_voff=voff, **options)
        self.voff, self.toff = voff, toff
        return self

    def __init__(self, pos, edges, alpha, beta, z, wgt, x0=None, tris=None, device=0,
                 _batch_voff=None, **options):
        self._lib = _l.load()
        self._h = C.c_void_p()
        pos = _f32(pos).reshape(-1, 2)
        edges = np.ascontiguousarray(edges, dtype=np.int32).reshape(-1, 2)
        alpha, beta, z, wgt = _f32(alpha), _f32(beta), _f32(z), _f32(wgt)
        self.V, self.E = pos.shape[0], edges.shape[0]
        if alpha.shape != (self.E,) or beta.shape != (self.E,) or z.shape != (self.V,) or \
                wgt.shape != (self.V,):
            raise ValueError("array shapes do not match V/E")
        x0 = None if x0 is None else _f32(x0)
        if tris is not None:
            tris = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
        self.T = 0 if tris is None else tris.shape[0]
        _l.check(self._lib.flame_hip_graph_create(C.byref(self._h), device, self.V, self.E, self.T),
                 "flame_hip_graph_create")
        try:
            for k, v in options.items():
                _l.check(self._lib.flame_hip_set_option(self._h, k.encode(), int(v)),
                         "flame_hip_set_option(%s)" % k)
            if _batch_voff is not None:
                vo = np.ascontiguousarray(_batch_voff, np.int32)
                _l.check(self._lib.flame_hip_graph_upload_batch(
                    self._h, len(vo) - 1, _ptr(vo), _ptr(pos), _ptr(edges), _ptr(alpha), _ptr(beta),
                    _ptr(z), _ptr(wgt), _ptr(x0), _ptr(tris)), "flame_hip_graph_upload_batch")
            else:
                _l.check(self._lib.flame_hip_graph_upload(self._h, _ptr(pos), _ptr(edges), _ptr(alpha),
                                                          _ptr(beta), _ptr(z), _ptr(wgt), _ptr(x0),
                                                          _ptr(tris)), "flame_hip_graph_upload")
        except Exception:
            self.close()
            raise

    @classmethod
    def empty(cls, device=0, **options):
        """A handle with no graph yet (frame streams: sync_features() / reupload() per frame)."""
        self = cls.__new__(cls)
        self._lib = _l.load()
        self._h = C.c_void_p()
        self.V = self.E = self.T = 0
        _l.check(self._lib.flame_hip_graph_create(C.byref(self._h), device, 0, 0, 0), "flame_hip_graph_create")
        for k, v in options.items():
            _l.check(self._lib.flame_hip_set_option(self._h, k.encode(), int(v)), "flame_hip_set_option(%s)" % k)
        return self

    def set_option(self, key, value):
        """flame_hip_set_option on a live handle (plan options take effect with the next upload; "persist" and
        "use_graph" with the next solve)."""
        _l.check(self._lib.flame_hip_set_option(self._h, key.encode(), int(value)), "flame_hip_set_option(%s)" % key)

    def sync_features(self, pos, idepth_mu, idepth_var, tris, sync_params, prediction=None):
        """Row a7 (graph sync): tracked features + their Delaunay triangulation -> edges, weights,
        data terms, initial x; resizes this handle and uploads.  Returns the data scale."""
        pos = _f32(pos).reshape(-1, 2)
        mu, var = _f32(idepth_mu), _f32(idepth_var)
        if isinstance(tris, (int, np.integer)):  # T of the list this handle's delaunay() made last: read where the library holds it
            T, tptr = int(tris), None
        else:
            tris = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
            T, tptr = len(tris), (_ptr(tris) if len(tris) else None)
        pred = None if prediction is None else _f32(prediction)
        scale = C.c_float()
        _l.check(self._lib.flame_hip_graph_sync(self._h, C.byref(sync_params), len(mu), T, _ptr(pos),
                                                _ptr(mu), _ptr(var), tptr,
                                                _ptr(pred), C.byref(scale)), "flame_hip_graph_sync")
        self.V, self.T, self.E = len(mu), T, self.info("E")
        return scale.value

    def delaunay(self, pos):
        """Row f3's first leg on the GPU (flame_hip_delaunay): counter-clockwise triangles of the Delaunay triangulation
        of pos (V x 2 pixel coordinates), each starting at its smallest vertex, ordered by that vertex."""
        pos = _f32(pos).reshape(-1, 2)
        tris = np.empty((max(2 * len(pos), 1), 3), np.int32)
        T = C.c_int32()
        _l.check(self._lib.flame_hip_delaunay(self._h, len(pos), _ptr(pos) if len(pos) else None, len(tris), _ptr(tris),
                                              C.byref(T)), "flame_hip_delaunay")
        return tris[:T.value].copy()

    def delaunay_keep(self, pos):
        """flame_hip_delaunay in KEEP mode: the list stays in the library (sync_features(..., tris=T, ...) reads it on the
        device), only T comes back; delaunay_list() hands the triangles out later."""
        pos = _f32(pos).reshape(-1, 2)
        T = C.c_int32()
        _l.check(self._lib.flame_hip_delaunay(self._h, len(pos), _ptr(pos) if len(pos) else None, 0, None, C.byref(T)),
                 "flame_hip_delaunay (keep)")
        self._kept_T = T.value
        return T.value

    def delaunay_list(self):
        tris = np.empty((max(getattr(self, "_kept_T", 0), 1), 3), np.int32)
        _l.check(self._lib.flame_hip_delaunay_list(self._h, len(tris), _ptr(tris)), "flame_hip_delaunay_list")
        return tris[:getattr(self, "_kept_T", 0)].copy()

    def edges(self):
        """The edge list derived by sync_features ([E,2] int32, i < j, lexicographic)."""
        e = np.empty((self.E, 2), np.int32)
        _l.check(self._lib.flame_hip_graph_edges(self._h, _ptr(e)), "flame_hip_graph_edges")
        return e

    def scale_state(self, s, sync=True):
        _l.check(self._lib.flame_hip_scale_state(self._h, float(s)), "flame_hip_scale_state")
        if sync:
            self.sync()

    def reupload(self, pos, edges, alpha, beta, z, wgt, x0=None, tris=None):
        """Next frame of a stream on the SAME handle (flame_hip_graph_resize + upload): a new
        graph of any size; stream, events, device buffers and plan buffers are kept."""
        pos = _f32(pos).reshape(-1, 2)
        edges = np.ascontiguousarray(edges, dtype=np.int32).reshape(-1, 2)
        alpha, beta, z, wgt = _f32(alpha), _f32(beta), _f32(z), _f32(wgt)
        V, E = pos.shape[0], edges.shape[0]
        if alpha.shape != (E,) or beta.shape != (E,) or z.shape != (V,) or wgt.shape != (V,):
            raise ValueError("array shapes do not match V/E")
        x0 = None if x0 is None else _f32(x0)
        if tris is not None:
            tris = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
        T = 0 if tris is None else tris.shape[0]
        _l.check(self._lib.flame_hip_graph_resize(self._h, V, E, T), "flame_hip_graph_resize")
        self.V, self.E, self.T = V, E, T
        _l.check(self._lib.flame_hip_graph_upload(self._h, _ptr(pos), _ptr(edges), _ptr(alpha),
                                                  _ptr(beta), _ptr(z), _ptr(wgt), _ptr(x0),
                                                  _ptr(tris)), "flame_hip_graph_upload")

    # -- lifetime --
    def close(self):
        if getattr(self, "_h", None):
            self._lib.flame_hip_graph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- info --
    def info(self, key):
        v = C.c_int64()
        _l.check(self._lib.flame_hip_get_info(self._h, key.encode(), C.byref(v)), "flame_hip_get_info")
        return v.value

    _PLAN_ELEM_BYTES = {"eij": 8, "t_eij": 8, "ew": 16, "t_ew": 16, "profile": 8, "tiles": C.sizeof(_l.TileDesc)}

    def plan_array(self, name, dtype):
        """Debug hook: copy of a host-side plan array (works on plan-only handles, device=-1)."""
        n = self._lib.flame_hip_debug_plan_array(self._h, name.encode(), None, 0)
        if n < 0:
            raise FlameHipError(int(n), "flame_hip_debug_plan_array")
        nbytes = int(n) * self._PLAN_ELEM_BYTES.get(name, 4)
        out = np.zeros(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        if n:
            self._lib.flame_hip_debug_plan_array(self._h, name.encode(), _ptr(out), out.nbytes)
        return out

    # -- the path --
    def set_state(self, x=None, w1=None, w2=None, xb=None, w1b=None, w2b=None, q=None):
        arrs = [None if a is None else _f32(a) for a in (x, w1, w2, xb, w1b, w2b, q)]
        _l.check(self._lib.flame_hip_set_state(self._h, *[_ptr(a) for a in arrs]), "flame_hip_set_state")

    def step(self, params, num_iters=1, stream=None, sync=True):
        """num_iters x (dualStep; primalStep; extraGradientStep)."""
        _l.check(self._lib.flame_hip_solve(self._h, C.byref(params), int(num_iters), stream),
                 "flame_hip_solve")
        if sync:
            self.sync()

    solve = step

    def sync(self):
        _l.check(self._lib.flame_hip_sync(self._h), "flame_hip_sync")

    def last_solve_ms(self):
        ms, n = C.c_float(), C.c_int32()
        _l.check(self._lib.flame_hip_last_solve_ms(self._h, C.byref(ms), C.byref(n)),
                 "flame_hip_last_solve_ms")
        return ms.value, n.value

    def costs(self, params):
        s, d = C.c_double(), C.c_double()
        _l.check(self._lib.flame_hip_costs(self._h, C.byref(params), C.byref(s), C.byref(d)),
                 "flame_hip_costs")
        return s.value, d.value

    def costs_masked(self, params, vmask=None, emask=None):
        """Costs over the flagged vertices / edges only (caller's order; None = all): what a
        multi-GPU subdomain owns."""
        vm = None if vmask is None else np.ascontiguousarray(vmask, np.uint8)
        em = None if emask is None else np.ascontiguousarray(emask, np.uint8)
        s, d = C.c_double(), C.c_double()
        _l.check(self._lib.flame_hip_costs_masked(self._h, C.byref(params), _ptr(vm), _ptr(em), C.byref(s),
                                                  C.byref(d)), "flame_hip_costs_masked")
        return s.value, d.value

    def smoothnessCost(self, params):
        return self.costs(params)[0]

    def dataCost(self, params):
        return self.costs(params)[1]

    def download(self, with_q=True):
        x, w1, w2 = (np.empty(self.V, np.float32) for _ in range(3))
        q = np.empty((self.E, 3), np.float32) if with_q else None
        _l.check(self._lib.flame_hip_download(self._h, _ptr(x), _ptr(w1), _ptr(w2), _ptr(q)),
                 "flame_hip_download")
        return x, w1, w2, q

    def download_bar(self):
        xb, w1b, w2b = (np.empty(self.V, np.float32) for _ in range(3))
        _l.check(self._lib.flame_hip_download_bar(self._h, _ptr(xb), _ptr(w1b), _ptr(w2b)),
                 "flame_hip_download_bar")
        return xb, w1b, w2b

    def triangles(self, Kinv, tri_params):
        Kinv = _f32(Kinv).reshape(9)
        vn = np.empty((self.V, 3), np.float32)
        tv = np.empty(self.T, np.uint8)
        tn = np.empty((self.T, 3), np.float32)
        _l.check(self._lib.flame_hip_triangles(self._h, _ptr(Kinv), C.byref(tri_params), _ptr(vn),
                                               _ptr(tv), _ptr(tn)), "flame_hip_triangles")
        return tn, tv, vn

    def frame_results(self, params, Kinv, tri_params, scale_back=1.0, with_edges=False, with_coverage=False):
        """What flame::Flame::update() reads back after the solve, with one synchronisation:
        (smooth, data, x[V], vtx_normals[V,3], tri_valid[T], edges[E,2] or None[, coverage])."""
        Kinv = _f32(Kinv).reshape(9)
        x = np.empty(self.V, np.float32)
        vn = np.empty((self.V, 3), np.float32)
        tv = np.empty(self.T, np.uint8)
        e = np.empty((self.E, 2), np.int32) if with_edges else None
        s, d = C.c_double(), C.c_double()
        cov = C.c_float()
        _l.check(self._lib.flame_hip_frame_results(self._h, C.byref(params), float(scale_back), _ptr(Kinv),
                                                   C.byref(tri_params), C.byref(s), C.byref(d), _ptr(x), _ptr(vn),
                                                   _ptr(tv), _ptr(e), C.byref(cov) if with_coverage else None),
                 "flame_hip_frame_results")
        if with_coverage:
            return s.value, d.value, x, vn, tv, e, cov.value
        return s.value, d.value, x, vn, tv, e

    def debug_image(self, kind, Kinv, tri_params, scene_color_scale=1.0, feat_pos=None, feat_mu=None):
        """Debug image of flame::Flame rendered on the device: BGR8 [H,W,3] (kind: lib.IMG_*)."""
        Kinv = _f32(Kinv).reshape(9)
        fp = None if feat_pos is None else _f32(feat_pos).reshape(-1, 2)
        fm = None if feat_mu is None else _f32(feat_mu)
        out = np.empty((tri_params.height, tri_params.width, 3), np.uint8)
        _l.check(self._lib.flame_hip_debug_image(self._h, int(kind), _ptr(Kinv), C.byref(tri_params),
                                                 float(scene_color_scale), 0 if fm is None else len(fm), _ptr(fp),
                                                 _ptr(fm), _ptr(out)), "flame_hip_debug_image")
        return out

    def mesh(self, Kinv, tri_params):
        """Row f1: (points[V,12] PointNormalUV layout, faces[F,3] reversed winding)."""
        Kinv = _f32(Kinv).reshape(9)
        pts = np.empty((self.V, 12), np.float32)
        faces = np.empty((max(self.T, 1), 3), np.int32)
        nf = C.c_int32()
        _l.check(self._lib.flame_hip_mesh(self._h, _ptr(Kinv), C.byref(tri_params), _ptr(pts),
                                          _ptr(faces), C.byref(nf)), "flame_hip_mesh")
        return pts, faces[:nf.value]

    def depthmaps(self, Kinv, tri_params, filtered=True, min_depth=0.1, max_depth=100.0, cloud=True):
        """Row f2: (idepthmap[H,W], depthmap[H,W], cloud[H,W,3] or None)."""
        Kinv = _f32(Kinv).reshape(9)
        H, W = tri_params.height, tri_params.width
        idm, dm = np.empty((H, W), np.float32), np.empty((H, W), np.float32)
        cl = np.empty((H, W, 3), np.float32) if cloud else None
        _l.check(self._lib.flame_hip_depthmaps(self._h, _ptr(Kinv), C.byref(tri_params), int(filtered),
                                               min_depth, max_depth, _ptr(idm), _ptr(dm), _ptr(cl)),
                 "flame_hip_depthmaps")
        return idm, dm, cl

    def graph_filter(self, kind, passes=1):
        """Row a9: median (kind 0) / low-pass (kind 1) filter of the vertex idepths."""
        _l.check(self._lib.flame_hip_graph_filter(self._h, int(kind), int(passes)), "flame_hip_graph_filter")
        self.sync()

    def update_data(self, z, wgt, x0=None):
        """New data terms on the same topology: state reset without rebuilding the plan."""
        z, wgt = _f32(z), _f32(wgt)
        x0 = None if x0 is None else _f32(x0)
        _l.check(self._lib.flame_hip_graph_update_data(self._h, _ptr(z), _ptr(wgt), _ptr(x0)),
                 "flame_hip_graph_update_data")
