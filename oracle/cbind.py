"""ctypes binding of oracle/libnltgv2_oracle.so (TEST INFRASTRUCTURE, see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def oracle_lib_path():
    return os.path.join(_HERE, "libnltgv2_oracle.so")


def build_oracle(force=False):
    """Compile the C restatement with the committed Makefile (gcc only, seconds)."""
    if force or not os.path.exists(oracle_lib_path()):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return oracle_lib_path()


class OracleParams(C.Structure):
    _fields_ = [("data_factor", C.c_float), ("step_x", C.c_float), ("step_q", C.c_float),
                ("theta", C.c_float), ("x_min", C.c_float), ("x_max", C.c_float)]


class _Graph(C.Structure):
    _fields_ = [("V", C.c_int32), ("E", C.c_int32),
                ("pos", C.c_void_p), ("edges", C.c_void_p), ("alpha", C.c_void_p),
                ("beta", C.c_void_p), ("z", C.c_void_p), ("wgt", C.c_void_p),
                ("x", C.c_void_p), ("w1", C.c_void_p), ("w2", C.c_void_p),
                ("xb", C.c_void_p), ("w1b", C.c_void_p), ("w2b", C.c_void_p),
                ("q", C.c_void_p)]


class TriParams(C.Structure):
    _fields_ = [("do_oblique_triangle_filter", C.c_int32), ("oblique_normal_thresh", C.c_float),
                ("oblique_idepth_diff_factor", C.c_float), ("oblique_idepth_diff_abs", C.c_float),
                ("do_edge_length_filter", C.c_int32), ("edge_length_thresh", C.c_float),
                ("do_idepth_triangle_filter", C.c_int32), ("min_triangle_idepth", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32)]


_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_oracle())
        _lib.nltgv2_solve.restype = C.c_int
    return _lib


def default_params(data_factor=0.15, step_x=1e-3, step_q=125.0, theta=0.25, x_min=0.0, x_max=10.0):
    """Defaults: reference cfg/flame_offline_tum.yaml:93-96; clamp 0..10 is upstream recall."""
    return OracleParams(data_factor, step_x, step_q, theta, x_min, x_max)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class COracle:
    """Holds one graph + solver state in NumPy arrays and steps it with the C restatement."""

    def __init__(self, pos, edges, alpha, beta, z, wgt, x0=None):
        self.pos = _f32(pos).reshape(-1, 2)
        self.edges = np.ascontiguousarray(edges, dtype=np.int32).reshape(-1, 2)
        self.V, self.E = self.pos.shape[0], self.edges.shape[0]
        self.alpha, self.beta, self.z, self.wgt = _f32(alpha), _f32(beta), _f32(z), _f32(wgt)
        assert self.alpha.shape == (self.E,) and self.beta.shape == (self.E,)
        assert self.z.shape == (self.V,) and self.wgt.shape == (self.V,)
        self.x = _f32(self.z if x0 is None else x0).copy()
        self.w1 = np.zeros(self.V, np.float32)
        self.w2 = np.zeros(self.V, np.float32)
        self.xb, self.w1b, self.w2b = self.x.copy(), self.w1.copy(), self.w2.copy()
        self.q = np.zeros((self.E, 3), np.float32)

    def set_state(self, x=None, w1=None, w2=None, xb=None, w1b=None, w2b=None, q=None):
        for name, val in dict(x=x, w1=w1, w2=w2, xb=xb, w1b=w1b, w2b=w2b).items():
            if val is not None:
                getattr(self, name)[:] = _f32(val)
        if q is not None:
            self.q[:] = _f32(q).reshape(self.E, 3)

    def _g(self):
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        return _Graph(self.V, self.E, p(self.pos), p(self.edges), p(self.alpha), p(self.beta),
                      p(self.z), p(self.wgt), p(self.x), p(self.w1), p(self.w2), p(self.xb),
                      p(self.w1b), p(self.w2b), p(self.q))

    def solve(self, params, num_iters):
        g = self._g()
        rc = _load().nltgv2_solve(C.byref(params), C.byref(g), int(num_iters))
        if rc != 0:
            raise MemoryError("nltgv2_solve")
        return self.x

    def solve_threads(self, params, num_iters, num_threads):
        """OpenMP variant (bit-identical to solve); used by bench.py's threaded cpu_baseline legs."""
        if not hasattr(self, "_row"):
            self._row = np.empty(self.V + 1, np.int32)
            self._inc = np.empty(2 * self.E, np.int32)
            g = self._g()
            _load().nltgv2_build_incidence(C.byref(g), self._row.ctypes.data_as(C.c_void_p),
                                           self._inc.ctypes.data_as(C.c_void_p))
        g = self._g()
        _load().nltgv2_solve_omp(C.byref(params), C.byref(g), self._row.ctypes.data_as(C.c_void_p),
                                 self._inc.ctypes.data_as(C.c_void_p), int(num_iters), int(num_threads))
        return self.x

    def graph_filter(self, kind):
        """Row a9: one Jacobi pass of the median (kind 0) or low-pass (kind 1) graph filter."""
        if not hasattr(self, "_row"):
            self._row = np.empty(self.V + 1, np.int32)
            self._inc = np.empty(2 * self.E, np.int32)
            g = self._g()
            _load().nltgv2_build_incidence(C.byref(g), self._row.ctypes.data_as(C.c_void_p),
                                           self._inc.ctypes.data_as(C.c_void_p))
        g = self._g()
        scratch = np.empty(max(self.V, 1), np.float32)
        _load().nltgv2_graph_filter(C.byref(g), self._row.ctypes.data_as(C.c_void_p),
                                    self._inc.ctypes.data_as(C.c_void_p), C.c_int32(kind),
                                    scratch.ctypes.data_as(C.c_void_p))

    def scale_state(self, s):
        """Row a7 epilogue: state and data term back to the caller's units (x *= s, ...)."""
        self.z = self.z.copy()  # may alias the caller's array
        g = self._g()
        _load().nltgv2_scale_state(C.byref(g), self.z.ctypes.data_as(C.c_void_p), C.c_float(s))

    def dual_step(self, params):
        g = self._g()
        _load().nltgv2_dual_step(C.byref(params), C.byref(g))

    def primal_step(self, params):
        g = self._g()
        xp, w1p, w2p = (np.empty(self.V, np.float32) for _ in range(3))
        vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        _load().nltgv2_primal_step(C.byref(params), C.byref(g), vp(xp), vp(w1p), vp(w2p))
        return xp, w1p, w2p

    def extragradient_step(self, params, xp, w1p, w2p):
        g = self._g()
        vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        _load().nltgv2_extragradient_step(C.byref(params), C.byref(g), vp(xp), vp(w1p), vp(w2p))

    def costs(self, params):
        g = self._g()
        s, d = C.c_double(), C.c_double()
        _load().nltgv2_costs(C.byref(params), C.byref(g), C.byref(s), C.byref(d))
        return s.value, d.value

    def apply_K(self, x, w1, w2):
        g = self._g()
        Ku = np.empty((self.E, 3), np.float32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        x, w1, w2 = _f32(x), _f32(w1), _f32(w2)
        _load().nltgv2_apply_K(C.byref(g), vp(x), vp(w1), vp(w2), vp(Ku))
        return Ku

    def apply_KT(self, q):
        g = self._g()
        q = _f32(q).reshape(self.E, 3)
        out = [np.empty(self.V, np.float32) for _ in range(3)]
        vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        _load().nltgv2_apply_KT(C.byref(g), vp(q), vp(out[0]), vp(out[1]), vp(out[2]))
        return out


class SyncParams(C.Structure):
    """nltgv2_sync_params (reference cfg/flame_offline_tum.yaml:89-92)."""
    _fields_ = [("adaptive_data_weights", C.c_int32), ("rescale_data", C.c_int32),
                ("init_with_prediction", C.c_int32), ("idepth_var_max_graph", C.c_float),
                ("edge_weight_rule", C.c_int32), ("alpha_gain", C.c_float), ("beta_gain", C.c_float)]


def feature_gate(var, var_max):
    var = _f32(var)
    keep = np.empty(len(var), np.uint8)
    L = _load()
    L.nltgv2_feature_gate.restype = C.c_int32
    n = L.nltgv2_feature_gate(C.c_int32(len(var)), var.ctypes.data_as(C.c_void_p), C.c_float(var_max),
                              keep.ctypes.data_as(C.c_void_p))
    assert n == int(keep.sum())
    return keep.astype(bool)


def graph_sync(sp, pos, mu, var, tris, prediction=None):
    """Row a7.  Returns dict(edges[E,2], alpha, beta, z, wgt, x0, scale)."""
    pos = _f32(pos).reshape(-1, 2)
    mu, var = _f32(mu), _f32(var)
    tris = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
    V, T = len(mu), len(tris)
    pred = None if prediction is None else _f32(prediction)
    edges = np.empty((3 * max(T, 1), 2), np.int32)
    alpha, beta = np.empty(3 * max(T, 1), np.float32), np.empty(3 * max(T, 1), np.float32)
    z, wgt, x0 = (np.empty(V, np.float32) for _ in range(3))
    scale = C.c_float()
    vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)  # noqa: E731
    L = _load()
    L.nltgv2_graph_sync.restype = C.c_int32
    E = L.nltgv2_graph_sync(C.byref(sp), C.c_int32(V), C.c_int32(T), vp(pos), vp(mu), vp(var), vp(tris),
                            vp(pred), vp(edges), vp(alpha), vp(beta), vp(z), vp(wgt), vp(x0),
                            C.byref(scale))
    return dict(edges=edges[:E].copy(), alpha=alpha[:E].copy(), beta=beta[:E].copy(), z=z, wgt=wgt,
                x0=x0, scale=scale.value)


def triangles(tri_params, Kinv, pos, x, tris):
    """Per-triangle stage (row a8).  Returns (tri_normals[T,3], tri_valid[T], vtx_normals[V,3])."""
    pos = _f32(pos).reshape(-1, 2)
    x = _f32(x)
    tris = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
    V, T = pos.shape[0], tris.shape[0]
    Kinv = _f32(Kinv).reshape(9)
    tn = np.empty((T, 3), np.float32)
    tv = np.empty(T, np.uint8)
    vn = np.empty((V, 3), np.float32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    _load().nltgv2_triangles(C.byref(tri_params), vp(Kinv), C.c_int32(V), C.c_int32(T), vp(pos),
                             vp(x), vp(tris), vp(tn), vp(tv), vp(vn))
    return tn, tv, vn


def mesh(Kinv, pos, x, vtx_normals, tris, tri_valid, width, height):
    """Row f1: (points[V,12] in PointNormalUV layout, faces[F,3] reversed winding)."""
    pos = _f32(pos).reshape(-1, 2)
    x, vn = _f32(x), _f32(vtx_normals).reshape(-1, 3)
    tris = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
    tv = np.ascontiguousarray(tri_valid, dtype=np.uint8)
    Kinv = _f32(Kinv).reshape(9)
    pts = np.empty((len(x), 12), np.float32)
    faces = np.empty((len(tris), 3), np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    L = _load()
    L.nltgv2_mesh_points(vp(Kinv), C.c_int32(len(x)), vp(pos), vp(x), vp(vn), C.c_int32(width),
                         C.c_int32(height), vp(pts))
    L.nltgv2_mesh_faces.restype = C.c_int32
    n = L.nltgv2_mesh_faces(C.c_int32(len(tris)), vp(tris), vp(tv), vp(faces))
    return pts, faces[:n]


def depthmaps(width, height, pos, x, tris, tri_valid, filtered, Kinv, min_depth, max_depth):
    """Row f2: (idepthmap[H,W], depthmap[H,W], cloud[H,W,3])."""
    pos = _f32(pos).reshape(-1, 2)
    x = _f32(x)
    tris = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
    tv = np.ascontiguousarray(tri_valid, dtype=np.uint8)
    Kinv = _f32(Kinv).reshape(9)
    idm = np.empty((height, width), np.float32)
    dm = np.empty((height, width), np.float32)
    cl = np.empty((height, width, 3), np.float32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    L = _load()
    L.nltgv2_idepthmap(C.c_int32(width), C.c_int32(height), C.c_int32(len(tris)), vp(pos), vp(x),
                       vp(tris), vp(tv), C.c_int32(int(filtered)), vp(idm))
    L.nltgv2_depth_and_cloud(C.c_int32(width), C.c_int32(height), vp(idm), vp(Kinv),
                             C.c_float(min_depth), C.c_float(max_depth), vp(dm), vp(cl))
    return idm, dm, cl


def coverage(idepthmap_filtered):
    """Stat key `coverage`: non-NaN share of the filtered dense idepthmap (float32)."""
    idm = _f32(idepthmap_filtered)
    H, W = idm.shape
    L = _load()
    L.nltgv2_coverage.restype = C.c_float
    return float(L.nltgv2_coverage(C.c_int32(W), C.c_int32(H), idm.ctypes.data_as(C.c_void_p)))


IMG_WIREFRAME, IMG_FEATURES, IMG_NORMALS, IMG_IDEPTHMAP = 0, 1, 2, 3


def debug_image(kind, width, height, scene_color_scale, pos, x, tris, tri_valid, vtx_normals=None,
                idepthmap_filtered=None, feat_pos=None, feat_mu=None):
    """The debug images of flame::Flame (BGR8 [H,W,3]); rules stated in nltgv2_oracle.c."""
    pos = _f32(pos).reshape(-1, 2)
    x = _f32(x)
    tris = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
    tv = np.ascontiguousarray(tri_valid, dtype=np.uint8)
    vn = None if vtx_normals is None else _f32(vtx_normals).reshape(-1, 3)
    idm = None if idepthmap_filtered is None else _f32(idepthmap_filtered)
    fp = None if feat_pos is None else _f32(feat_pos).reshape(-1, 2)
    fm = None if feat_mu is None else _f32(feat_mu)
    out = np.empty((height, width, 3), np.uint8)
    vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)  # noqa: E731
    _load().nltgv2_debug_image(C.c_int32(kind), C.c_int32(width), C.c_int32(height), C.c_float(scene_color_scale),
                               C.c_int32(len(x)), vp(pos), vp(x), C.c_int32(len(tris)), vp(tris), vp(tv), vp(vn),
                               vp(idm), C.c_int32(0 if fm is None else len(fm)), vp(fp), vp(fm), vp(out))
    return out
