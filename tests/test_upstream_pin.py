"""The receiving end of an upstream pin.

Parity against upstream is UNPINNED today: robustrobotics/flame is not in the reference tree and
cannot be built here.  A maintainer with a real build runs tools/pin_upstream/dump_upstream.cc
(upstream's own step() on a scene of this repository), converts the dump with
tools/pin_upstream/convert_dump.py into tests/golden/upstream_<tag>.npz, and these tests then run
the oracle (CPU) and the HIP path (-m gpu) against it:

* every tests/golden/upstream_*.npz present is checked (none present -> those tests skip);
* the whole pipeline (scene writer -> dump program with the shared fldump.h writer -> converter ->
  checker) is exercised with a self-generated STAND-IN dump (tests/cpp/pin_standin_dump.cc: the
  oracle playing upstream), including the [UPSTREAM-RECALL] switches a mismatch would flip
  (d_sign, idepth clamp).  A stand-in dump pins nothing; it proves the receiving end works.

Tolerance: the north_star's 1e-4 RMS on the idepths; bit-exactness is reported, not required (an
upstream that sums in another order may differ in the last bits; DESIGN.md "Oracle")."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import COracle
from oracle.cbind import default_params as oracle_params
from tests.util import graphgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "pin_upstream"))
import convert_dump  # noqa: E402
import make_scene  # noqa: E402

TOL_RMS = 1e-4
UPSTREAM = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "upstream_*.npz")))


def load_dump(path):
    d = dict(np.load(path, allow_pickle=False))
    d["d_sign"] = int(d["d_sign"]) if "d_sign" in d else 1
    return d


def oracle_states(d):
    """The oracle on the dump's inputs with the dump's switches: {n: (x, w1, w2, q)}."""
    pos = d["pos"] * np.float32(d["d_sign"])  # the oracle uses pos only through d = pos_i - pos_j
    o = COracle(pos, d["edges"], d["alpha"], d["beta"], d["z"], d["wgt"], x0=d["x0"])
    p = oracle_params(*[float(v) for v in d["params"]])
    out, done = {}, 0
    for n in [int(v) for v in d["iters"]]:
        o.solve(p, n - done)
        done = n
        out[n] = (o.x.copy(), o.w1.copy(), o.w2.copy(), o.q.copy())
    return out


def compare(d, states, what):
    rep = []
    for n, (x, w1, w2, q) in states.items():
        ref = d["x_after_%d" % n]
        rms = float(np.sqrt(np.mean((x.astype(np.float64) - ref) ** 2)))
        exact = bool(np.array_equal(x.view(np.uint32), ref.view(np.uint32)))
        rep.append((n, rms, exact))
        assert rms <= TOL_RMS, "%s: idepth RMS %.3e after %d iterations (tolerance %.0e)" % (what, rms, n, TOL_RMS)
        for name, got in (("w1", w1), ("w2", w2)):
            if name + "_after_%d" % n in d:
                r = d[name + "_after_%d" % n]
                assert float(np.sqrt(np.mean((got.astype(np.float64) - r) ** 2))) <= TOL_RMS, (what, name, n)
        if "q_after_%d" % n in d:
            assert float(np.abs(q - d["q_after_%d" % n]).max()) <= 1e-2, (what, "q", n)
    return rep


@pytest.fixture(scope="module")
def standin_exe(tmp_path_factory, oracle_built):
    out = str(tmp_path_factory.mktemp("pin") / "pin_standin_dump")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-Werror",
                           os.path.join(ROOT, "tests", "cpp", "pin_standin_dump.cc"), "-o", out,
                           "-L" + os.path.join(ROOT, "oracle"), "-lnltgv2_oracle",
                           "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    return out


def make_standin(exe, tmp_path, d_sign=1, x_max=10.0, iters=(1, 10, 60), V=400):
    g = graphgen.synthetic(V, seed=11)
    scene, dump, npz = (str(tmp_path / n) for n in ("scene.txt", "s.fldump", "standin.npz"))
    make_scene.write_scene(scene, g, x0=g.z * np.float32(0.9))
    subprocess.check_call([exe, scene, dump, str(d_sign), str(x_max)] + [str(n) for n in iters])
    sys.argv = ["convert_dump", dump, npz, "d_sign=%d" % d_sign, "source=stand-in (oracle)"]
    convert_dump.main()
    return npz


@pytest.mark.parametrize("d_sign,x_max", [(1, 10.0), (-1, 10.0), (1, 0.6)])
def test_pin_pipeline_with_standin_dump(standin_exe, tmp_path, d_sign, x_max):
    d = load_dump(make_standin(standin_exe, tmp_path, d_sign, x_max))
    assert d["d_sign"] == d_sign and abs(float(d["params"][5]) - x_max) < 1e-6
    rep = compare(d, oracle_states(d), "oracle vs stand-in")
    assert all(exact for _, _, exact in rep)  # the stand-in IS the oracle
    if x_max < 1.0:  # the clamp matters: ignoring the recorded switch must be visible
        wrong = dict(d, params=np.array([0.15, 1e-3, 125.0, 0.25, 0.0, 10.0], np.float32))
        assert not np.array_equal(oracle_states(wrong)[60][0], d["x_after_60"])
    if d_sign < 0:
        # K10 (gauge symmetry): d -> -d maps (x, w, q1, q2, q3) to (x, -w, q1, -q2, -q3) exactly
        # (negation is exact in float32), so from w = 0, q = 0 the IDEPTHS do not depend on the sign
        # convention of the edge vector at all -- one [UPSTREAM-RECALL] item that cannot break parity
        # of x; the switch only matters when plane slopes or duals are compared.
        x, w1, w2, q = oracle_states(dict(d, d_sign=1))[60]
        assert np.array_equal(x, d["x_after_60"])
        assert np.array_equal(w1, -d["w1_after_60"]) and np.array_equal(w2, -d["w2_after_60"])
        assert np.array_equal(q[:, 0], d["q_after_60"][:, 0]) and np.array_equal(q[:, 1:], -d["q_after_60"][:, 1:])
        assert np.abs(w1).max() > 0


@pytest.mark.parametrize("path", UPSTREAM or [None])
def test_oracle_against_upstream_dump(path):
    if path is None:
        pytest.skip("no tests/golden/upstream_*.npz: parity vs upstream is unpinned (tools/pin_upstream/README.md)")
    d = load_dump(path)
    for n, rms, exact in compare(d, oracle_states(d), "oracle vs %s" % os.path.basename(path)):
        print("%s: after %d iterations RMS %.3e, bit-exact %s" % (os.path.basename(path), n, rms, exact))


def hip_states(d):
    from flame_ros_amd.regularizer import GraphRegularizer
    from flame_ros_amd.lib import Params
    p = Params(*[float(v) for v in d["params"]])
    out, done = {}, 0
    with GraphRegularizer(d["pos"], d["edges"], d["alpha"], d["beta"], d["z"], d["wgt"], x0=d["x0"],
                          d_sign=d["d_sign"]) as r:
        for n in [int(v) for v in d["iters"]]:
            r.step(p, n - done)
            done = n
            x, w1, w2, q = r.download()
            out[n] = (x, w1, w2, q)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("d_sign,x_max", [(1, 10.0), (-1, 10.0), (1, 0.6)])
def test_hip_against_standin_dump(gpu, standin_exe, tmp_path, d_sign, x_max):
    d = load_dump(make_standin(standin_exe, tmp_path, d_sign, x_max, V=3000))
    rep = compare(d, hip_states(d), "HIP vs stand-in")
    assert all(exact for _, _, exact in rep)  # the arithmetic contract: the oracle's bits


@pytest.mark.gpu
@pytest.mark.parametrize("path", UPSTREAM or [None])
def test_hip_against_upstream_dump(gpu, path):
    if path is None:
        pytest.skip("no tests/golden/upstream_*.npz: parity vs upstream is unpinned (tools/pin_upstream/README.md)")
    d = load_dump(path)
    compare(d, hip_states(d), "HIP vs %s" % os.path.basename(path))


# ---------------------------------------------------------------------------------------------------------------------
# What LEAVES the boundary: tools/pin_upstream/dump_upstream_frame.cc records upstream's
# getRawIDepths / getInverseDepthMesh / getFilteredInverseDepthMap (reference src/flame_offline_tum.cc:628-643, 680-682)
# after a real Flame::update; these checks feed upstream's vertices + triangles + raw idepths through graph sync -> solve
# -> per-triangle stage -> dense map of the oracle (CPU) and of the HIP path (-m gpu) and compare rows a7, a8, f2 and the
# mesh idepths.  None exists yet (tests skip); the pipeline is exercised with a stand-in frame dump written in the same
# container format (the oracle playing upstream).
FRAMES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "upstream_frame_*.npz")))
FRAME_ITERS_DEFAULT = 200  # upstream's iterations per update are not exposed by any flame_ros key: key `iters` of the npz


def write_fldump(path, arrays):
    """FLDUMP1 (tools/pin_upstream/fldump.h) from Python: what the stand-in uses instead of the C++ writer."""
    import struct
    with open(path, "wb") as f:
        f.write(b"FLDUMP1\n")
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            dt = b"f" if a.dtype == np.float32 else b"i"
            assert a.dtype in (np.float32, np.int32), name
            f.write(name.encode() + b"\n" + dt + struct.pack("<I", a.ndim) + struct.pack("<%dI" % a.ndim, *a.shape) + a.tobytes())


def _tri_params(d, cls):
    on, tf = [int(v) for v in d["tri_filter_on"]], [float(v) for v in d["tri_filter"]]
    W, H = [int(v) for v in d["image_size"]]
    return cls(on[0], tf[0], tf[1], tf[2], on[1], tf[3], on[2], tf[4], W, H)


def frame_inputs(d):
    """upstream's raw features through the variance gate (row a7): the vertices, in order, must be upstream's mesh vertices"""
    from oracle.cbind import feature_gate
    keep = feature_gate(d["raw_var"], float(d["sync"][0]))
    pos, mu, var = d["raw_pos"][keep], d["raw_mu"][keep], d["raw_var"][keep]
    assert pos.shape == d["mesh_pos"].shape and np.array_equal(pos, d["mesh_pos"]), "the gated features are not upstream's mesh vertices"
    iters = int(d["iters"]) if "iters" in d else FRAME_ITERS_DEFAULT
    return pos, mu, var, d["mesh_tris"].astype(np.int32), iters


def frame_oracle(d):
    from oracle.cbind import SyncParams as OSync, TriParams as OTri, graph_sync as oracle_sync, triangles as otri, depthmaps as odm
    pos, mu, var, tris, iters = frame_inputs(d)
    s = oracle_sync(OSync(int(d["sync"][1]), int(d["sync"][2]), int(d["sync"][3]), float(d["sync"][0])), pos, mu, var, tris, None)
    o = COracle(pos, s["edges"], s["alpha"], s["beta"], s["z"], s["wgt"], x0=s["x0"])
    rp = [float(v) for v in d["rparams"]]
    o.solve(oracle_params(rp[0], rp[1], rp[2], rp[3]), iters)
    x = o.x * np.float32(s["scale"])
    Kinv = np.linalg.inv(d["K"].astype(np.float64)).astype(np.float32)
    tp = _tri_params(d, OTri)
    tn, tv, vn = otri(tp, Kinv, pos, x, tris)
    idm = odm(tp.width, tp.height, pos, x, tris, tv, True, Kinv, 0.1, 100.0)[0]
    return dict(x=x, normals=vn, tri_valid=tv, edges=s["edges"], idepthmap=idm)


def frame_hip(d):
    from flame_ros_amd.lib import Params, TriParams
    from flame_ros_amd.regularizer import GraphRegularizer, default_sync_params
    pos, mu, var, tris, iters = frame_inputs(d)
    sp = default_sync_params(bool(d["sync"][1]), bool(d["sync"][2]), bool(d["sync"][3]), float(d["sync"][0]))
    rp = [float(v) for v in d["rparams"]]
    p = Params(rp[0], rp[1], rp[2], rp[3], 0.0, 10.0)
    Kinv = np.linalg.inv(d["K"].astype(np.float64)).astype(np.float32)
    tp = _tri_params(d, TriParams)
    with GraphRegularizer.empty(device=0) as r:
        scale = r.sync_features(pos, mu, var, tris, sp)
        r.step(p, iters, sync=False)
        _, _, x, vn, tv, edges = r.frame_results(p, Kinv, tp, scale_back=scale, with_edges=True)[:6]
        idm = r.depthmaps(Kinv, tp, filtered=True, cloud=False)[0]
    return dict(x=x, normals=vn, tri_valid=tv, edges=edges, idepthmap=idm)


def compare_frame(d, got, what, exact=False):
    x = d["mesh_idepth"]
    rms = float(np.sqrt(np.nanmean((got["x"].astype(np.float64) - x) ** 2)))
    assert rms <= TOL_RMS, "%s: mesh idepth RMS %.3e (tolerance %.0e)" % (what, rms, TOL_RMS)
    canon = lambda e: set(map(tuple, np.sort(np.asarray(e).reshape(-1, 2), axis=1).tolist()))  # noqa: E731
    assert canon(got["edges"]) == canon(d["mesh_edges"]), "%s: edge sets differ" % what
    tv_ref = d["mesh_tri_valid"].astype(bool)
    agree = float(np.mean(got["tri_valid"].astype(bool) == tv_ref))
    assert agree >= (1.0 if exact else 0.995), "%s: triangle validity agrees on %.4f of the triangles" % (what, agree)
    n_ref, n_got = d["mesh_normals"], got["normals"]
    ok = np.isfinite(n_ref).all(1) & np.isfinite(n_got).all(1)
    assert ok.mean() > 0.99 and float(np.abs(n_ref[ok] - n_got[ok]).max()) <= (0.0 if exact else 1e-3), "%s: vertex normals" % what
    m_ref, m_got = d["idepthmap_filtered"], got["idepthmap"]
    both = np.isfinite(m_ref) & np.isfinite(m_got)
    same_mask = float(np.mean(np.isfinite(m_ref) == np.isfinite(m_got)))
    assert same_mask >= (1.0 if exact else 0.995), "%s: filtered map covers the same pixels on %.4f of the image" % (what, same_mask)
    if both.any():
        mrms = float(np.sqrt(np.mean((m_ref[both].astype(np.float64) - m_got[both]) ** 2)))
        assert mrms <= TOL_RMS, "%s: filtered idepthmap RMS %.3e" % (what, mrms)
    if exact:
        assert np.array_equal(got["x"].view(np.uint32), x.view(np.uint32)), what
    return rms


def make_frame_standin(tmp_path, V=1500, iters=60):
    """a frame dump with the oracle playing upstream, through the real container + converter"""
    from oracle.cbind import SyncParams as OSync, TriParams as OTri, graph_sync as oracle_sync, triangles as otri, depthmaps as odm
    g = graphgen.synthetic(V, seed=17)
    rng = np.random.default_rng(3)
    extra = 40  # features that fail the variance gate (they follow the vertices: the gate keeps order)
    raw_pos = np.concatenate([g.pos, (rng.random((extra, 2)) * [g.width, g.height]).astype(np.float32)])
    raw_mu = np.concatenate([g.z, np.full(extra, 0.7, np.float32)])
    raw_var = np.concatenate([np.full(g.V, 1e-4, np.float32), np.full(extra, 0.5, np.float32)])
    K = np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]], np.float32)
    Kinv = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
    s = oracle_sync(OSync(0, 0, 1, 0.01), g.pos, g.z, raw_var[:g.V], g.tris, None)
    o = COracle(g.pos, s["edges"], s["alpha"], s["beta"], s["z"], s["wgt"], x0=s["x0"])
    o.solve(oracle_params(), iters)
    tp = OTri(1, 1.57, 0.35, 0.1, 1, 0.333, 1, 0.01, g.width, g.height)
    tn, tv, vn = otri(tp, Kinv, g.pos, o.x, g.tris)
    idm = odm(g.width, g.height, g.pos, o.x, g.tris, tv, True, Kinv, 0.1, 100.0)[0]
    dump, npz = str(tmp_path / "frame.fldump"), str(tmp_path / "frame_standin.npz")
    write_fldump(dump, dict(
        image_size=np.array([g.width, g.height], np.int32), K=K, rparams=np.array([0.15, 1e-3, 125.0, 0.25], np.float32),
        tri_filter=np.array([1.57, 0.35, 0.1, 0.333, 0.01], np.float32), tri_filter_on=np.array([1, 1, 1], np.int32),
        sync=np.array([0.01, 0, 0, 1], np.float32), raw_pos=raw_pos, raw_mu=raw_mu, raw_var=raw_var, mesh_pos=g.pos,
        mesh_idepth=o.x, mesh_normals=vn, mesh_tris=g.tris.astype(np.int32), mesh_tri_valid=tv.astype(np.int32),
        mesh_edges=s["edges"].astype(np.int32), idepthmap_filtered=idm))
    sys.argv = ["convert_dump", dump, npz, "iters=%d" % iters, "source=stand-in (oracle)"]
    convert_dump.main()
    return npz


def test_frame_pin_pipeline_with_standin(tmp_path, oracle_built):
    d = dict(np.load(make_frame_standin(tmp_path), allow_pickle=False))
    assert int(d["iters"]) == 60 and len(d["raw_mu"]) == len(d["mesh_idepth"]) + 40
    compare_frame(d, frame_oracle(d), "oracle vs stand-in frame", exact=True)
    wrong = dict(d, iters=np.array(40))  # the recorded iteration count matters: ignoring it must be visible
    with pytest.raises(AssertionError):
        compare_frame(wrong, frame_oracle(wrong), "wrong iteration count", exact=True)


@pytest.mark.parametrize("path", FRAMES or [None])
def test_oracle_against_upstream_frame_dump(path, oracle_built):
    if path is None:
        pytest.skip("no tests/golden/upstream_frame_*.npz: rows a7 / a8 / f2 are unpinned (tools/pin_upstream/README.md)")
    d = dict(np.load(path, allow_pickle=False))
    print("%s: mesh idepth RMS %.3e" % (os.path.basename(path), compare_frame(d, frame_oracle(d), "oracle vs " + os.path.basename(path))))


@pytest.mark.gpu
def test_hip_against_standin_frame_dump(gpu, tmp_path, oracle_built):
    d = dict(np.load(make_frame_standin(tmp_path, V=4000), allow_pickle=False))
    compare_frame(d, frame_hip(d), "HIP vs stand-in frame", exact=True)  # (the arithmetic contract: the oracle's bits)


@pytest.mark.gpu
@pytest.mark.parametrize("path", FRAMES or [None])
def test_hip_against_upstream_frame_dump(gpu, path):
    if path is None:
        pytest.skip("no tests/golden/upstream_frame_*.npz: rows a7 / a8 / f2 are unpinned (tools/pin_upstream/README.md)")
    d = dict(np.load(path, allow_pickle=False))
    compare_frame(d, frame_hip(d), "HIP vs " + os.path.basename(path))
