"""The C++ facade include/flame/flame.h (flame::Flame reduced to the regulariser path) compiled as
C++11 (the reference's standard, reference CMakeLists.txt:26) and driven the way flame_ros does.
CPU: it must compile, link against libflame_hip.so, and updateGraph() must return false without a
GPU (reference error convention: caller warns and skips the frame, src/flame_offline_tum.cc:
597-601).  GPU: its output equals the oracle bit for bit."""
import os
import struct
import subprocess

import numpy as np
import pytest

from flame_ros_amd import lib
from oracle import COracle
from oracle.cbind import default_params, triangles as oracle_triangles, TriParams
from tests.util import graphgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    lib.load()
    out = str(tmp_path_factory.mktemp("facade") / "facade_conformance")
    cmd = ["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "facade_conformance.cc"), "-o", out,
           "-L" + os.path.join(ROOT, "flame_ros_amd"), "-lflame_hip",
           "-Wl,-rpath," + os.path.join(ROOT, "flame_ros_amd"), "-pthread"]
    subprocess.check_call(cmd)
    return out


def write_input(path, g, iters, device):
    with open(path, "wb") as f:
        f.write(struct.pack("<4i", g.V, g.T, iters, device))
        f.write(g.pos.astype(np.float32).tobytes())
        f.write(g.z.astype(np.float32).tobytes())
        f.write(g.tris.astype(np.int32).tobytes())


def test_facade_compiles_and_fails_cleanly_without_gpu(exe, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    g = graphgen.synthetic(500, seed=1)
    inp = str(tmp_path / "in.bin")
    write_input(inp, g, 10, 0)
    p = subprocess.run([exe, inp, str(tmp_path / "o.bin"), str(tmp_path / "o.txt")],
                       capture_output=True, text=True)
    assert p.returncode == 3, (p.returncode, p.stdout, p.stderr)
    assert "update=0" in p.stdout and "hip_error=%d" % lib.ERR_NODEVICE in p.stdout


@pytest.mark.gpu
def test_facade_matches_oracle_on_gpu(gpu, exe, tmp_path):
    g = graphgen.synthetic(3000, seed=2)
    iters = 120
    inp, ob, ot = (str(tmp_path / n) for n in ("in.bin", "o.bin", "o.txt"))
    write_input(inp, g, iters, 0)
    p = subprocess.run([exe, inp, ob, ot], capture_output=True, text=True)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    raw = open(ob, "rb").read()
    x = np.frombuffer(raw[:4 * g.V], np.float32)
    vn = np.frombuffer(raw[4 * g.V:16 * g.V], np.float32).reshape(-1, 3)
    tv = np.frombuffer(raw[16 * g.V:], np.uint8)
    nE, smooth, data, avg_smooth, upd_ms = open(ot).read().split()
    o = COracle(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt)
    po = default_params()
    o.solve(po, iters)
    assert int(nE) == g.E
    assert np.array_equal(x.view(np.uint32), o.x.view(np.uint32))
    so, do = o.costs(po)
    assert abs(float(smooth) - so) <= 1e-9 * so and abs(float(data) - do) <= 1e-9 * do
    assert abs(float(avg_smooth) - so / g.V) <= 1e-9 * so and float(upd_ms) > 0
    Kinv = np.array([[1 / 525, 0, -319.5 / 525], [0, 1 / 525, -239.5 / 525], [0, 0, 1]], np.float32)
    Kinv = np.array([[np.float32(1) / np.float32(525), 0, np.float32(-319.5) / np.float32(525)],
                     [0, np.float32(1) / np.float32(525), np.float32(-239.5) / np.float32(525)],
                     [0, 0, 1]], np.float32)
    tp = TriParams(1, 1.57, 0.35, 0.1, 1, 0.333, 1, 0.01, 640, 480)
    _, tv_o, vn_o = oracle_triangles(tp, Kinv, g.pos, o.x, g.tris)
    assert np.array_equal(tv, tv_o)
    assert np.array_equal(vn.view(np.uint32), vn_o.view(np.uint32))
