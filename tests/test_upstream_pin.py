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
