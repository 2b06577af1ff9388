"""-m gpu: BASELINE config 1's and config 3's plumbing end to end without ROS -- a TUM-format sequence (index file,
8-bit RGB PNGs, 16-bit depth PNGs) and an ASL-format one (pose / cam0 / depth0 sensor folders with sensor.yaml +
data.csv, a distorted colour camera) of a known two-plane scene with depth holes, through tools/flame_offline_lite.cc:
dataset index -> PNG decode (-> rectification) -> flame::Flame::update() with a FrontEnd (grid features with idepth from
the depth image: the stand-in for upstream's feature pipeline) -> idepth mesh, stats.

Every frame must come back ok, and what the regulariser returned must be the ORACLE's bits on what went in: the tool
dumps each frame's gated features + triangles + mesh idepths, the test runs oracle graph sync + 200 iterations on the
same features.  Reference flow: src/flame_offline_tum.cc:565-707, src/flame_offline_asl.cc:398-435 (pose: doubles cast
to float), src/ros_sensor_streams/asl_rgbd_offline_stream.cc:152-345."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H = 640, 480


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("fol") / "flame_offline_lite")
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "flame_offline_lite.cc"), "-o", out,
                           "-L" + os.path.join(ROOT, "flame_ros_amd"), "-lflame_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "flame_ros_amd"), "-pthread"])
    return out


def scene(k, rng):
    """Two fronto-parallel-ish planes: idepth affine in (u, v), a step near the image centre; depth holes."""
    vv, uu = np.mgrid[0:H, 0:W]
    idepth = 0.45 + 0.0004 * uu - 0.0002 * vv + 0.25 * (uu > W // 2 + 10 * k)
    rgb = np.clip((idepth[..., None] * 160 + rng.normal(0, 6, (H, W, 3))), 0, 255).astype(np.uint8)
    return idepth, rgb, None


def check_frames_against_oracle(dump_dir, rows, iters):
    """The mesh idepths of every dumped frame = oracle graph sync + `iters` iterations on the dumped features."""
    from oracle import COracle
    from oracle.cbind import SyncParams as OSync, default_params as oparams, graph_sync as oracle_sync
    n = 0
    for r in rows:
        f = os.path.join(dump_dir, "frame_%s.bin" % r["frame"])
        raw = open(f, "rb").read()
        V, T = np.frombuffer(raw, np.int32, 2)
        off = 8
        pos = np.frombuffer(raw, np.float32, 2 * V, off).reshape(V, 2); off += 8 * V
        mu = np.frombuffer(raw, np.float32, V, off); off += 4 * V
        var = np.frombuffer(raw, np.float32, V, off); off += 4 * V
        tris = np.frombuffer(raw, np.int32, 3 * T, off).reshape(T, 3); off += 12 * T
        x = np.frombuffer(raw, np.float32, V, off)
        assert V == int(r["vtx"]) and T == int(r["tris"])
        # cfg/flame_offline_tum.yaml:87-92 defaults: no adaptive weights, no rescale, prediction init (none given), gate 0.01
        s = oracle_sync(OSync(0, 0, 1, 0.01), pos.copy(), mu.copy(), var.copy(), tris.copy(), None)
        assert len(s["edges"]) == int(r["edges"])
        o = COracle(pos.copy(), s["edges"], s["alpha"], s["beta"], s["z"], s["wgt"], x0=s["x0"])
        o.solve(oparams(), iters)
        assert np.array_equal(x.view(np.uint32), o.x.view(np.uint32)), "frame %s: mesh idepths differ from the oracle" % r["frame"]
        so, do = o.costs(oparams())
        assert abs(float(r["cost_smooth"]) - so) <= 1e-5 * so and abs(float(r["cost_data"]) - do) <= 1e-5 * max(do, 1e-9)
        n += 1
    return n


def parse(stdout):
    return [dict(zip(l.split()[0::2], l.split()[1::2])) for l in stdout.splitlines() if l.startswith("frame")]


def common_checks(rows, nframes):
    assert len(rows) == nframes
    for r in rows:
        assert r["ok"] == "1" and r["hip_error"] == "0"
        assert 1000 <= int(r["vtx"]) <= 1200 and int(r["feats"]) == int(r["vtx"])  # 40 x 30 cells minus the holes
        assert int(r["tris"]) > 1800 and int(r["edges"]) > 2800
        assert 0.6 < float(r["coverage"]) <= 1.0
        assert float(r["rms_vs_truth"]) < 0.03, r  # (sanity only; the bits are checked against the oracle)
    # (the first update creates the GPU context, streams and buffers; the later ones are a matter of a millisecond: the
    # best of them shows that, the worst only has to be bounded -- the pool's hosts are shared, and an update() is a
    # handful of host threads that can be held up by whatever else runs there)
    later = [float(r["update_ms"]) for r in rows[1:]]
    assert min(later) < 5.0 and max(later) < 2000.0, later


def test_tum_sequence_through_the_facade(gpu, exe, tmp_path):
    PIL = pytest.importorskip("PIL.Image")
    seq = tmp_path / "seq"
    (seq / "rgb").mkdir(parents=True)
    (seq / "depth").mkdir()
    (tmp_path / "dump").mkdir()
    fx, fy, cx, cy = 525.0, 525.0, 319.5, 239.5
    rng = np.random.default_rng(0)
    lines = ["# synthetic sequence"]
    for k in range(4):
        idepth, rgb, _ = scene(k, rng)
        raw = np.round(5000.0 / idepth).astype(np.uint16)
        raw[rng.random((H, W)) < 0.03] = 0           # holes: no measurement (no feature there)
        raw[40:90, 500:600] = 0
        PIL.fromarray(rgb, mode="RGB").save(str(seq / "rgb" / ("%d.png" % k)))
        PIL.fromarray(raw).save(str(seq / "depth" / ("%d.png" % k)))
        t = 1305031102.175304 + 0.033 * k
        lines.append("%.6f 1.34 0.62 1.65 0.6574 0.6126 -0.2949 -0.3248 %.6f rgb/%d.png %.6f depth/%d.png" % (t, t, k, t, k))
    (seq / "index.txt").write_text("\n".join(lines) + "\n")
    p = subprocess.run([exe, str(seq / "index.txt"), "RDF", str(fx), str(fy), str(cx), str(cy), "200", "--dump", str(tmp_path / "dump")],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    rows = parse(p.stdout)
    common_checks(rows, 4)
    assert check_frames_against_oracle(str(tmp_path / "dump"), rows, 200) == 4
    # a missing image file is reported, not a crash
    os.remove(str(seq / "rgb" / "2.png"))
    p = subprocess.run([exe, "tum", str(seq / "index.txt"), "RDF", str(fx), str(fy), str(cx), str(cy)], capture_output=True, text=True)
    assert p.returncode == 4 and "cannot read" in p.stderr


SENSOR_POSE = """sensor_type: pose
T_BS:
  cols: 4
  rows: 4
  data: [1.0, 0.0, 0.0, 0.02,
         0.0, 1.0, 0.0, -0.01,
         0.0, 0.0, 1.0, 0.05,
         0.0, 0.0, 0.0, 1.0]
"""
SENSOR_CAM = """sensor_type: camera
rate_hz: 20
resolution: [640, 480]
camera_model: pinhole
intrinsics: [458.654, 457.296, 327.127, 238.253]
distortion_model: radial-tangential
distortion_coefficients: [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05]
T_BS:
  cols: 4
  rows: 4
  data: [0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975,
         0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768,
         -0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949,
         0.0, 0.0, 0.0, 1.0]
"""
SENSOR_DEPTH = "sensor_type: depth\ndepth_scale_factor: 1000.0\n"


def test_asl_sequence_through_the_facade(gpu, exe, tmp_path):
    """BASELINE config 3's plumbing: EuRoC-style sensor folders (K, radial-tangential distortion, depth scale and the
    body transforms from sensor.yaml; 100 Hz poses associated with 20 Hz images) -> rectified gray image ->
    update(), the pose handed over as the reference does: computed in double precision, cast to float at the
    Sophus::SE3f (src/flame_offline_asl.cc:431)."""
    PIL = pytest.importorskip("PIL.Image")
    from scipy.spatial.transform import Rotation as R
    root = tmp_path / "asl"
    for name, yaml in (("pose", SENSOR_POSE), ("cam0", SENSOR_CAM), ("depth0", SENSOR_DEPTH)):
        (root / name / "data").mkdir(parents=True)
        (root / name / "sensor.yaml").write_text(yaml)
    (tmp_path / "dump").mkdir()
    rng = np.random.default_rng(1)
    t0 = 1403715273262142976
    pose_t = [t0 + k * 10_000_000 for k in range(25)]
    img_t = [t0 + 2_000_000 + k * 50_000_000 for k in range(4)]
    poses = []
    with open(root / "pose" / "data.csv", "w") as f:
        f.write("#timestamp [ns],p_x,p_y,p_z,q_w,q_x,q_y,q_z\n")
        for t in pose_t:
            q = R.from_rotvec(rng.normal(0, 0.3, 3)).as_quat()  # x y z w
            p = rng.normal(0, 1, 3)
            poses.append((t, p, q))
            f.write("%d,%.12f,%.12f,%.12f,%.12f,%.12f,%.12f,%.12f\n" % (t, p[0], p[1], p[2], q[3], q[0], q[1], q[2]))
    for name in ("cam0", "depth0"):
        with open(root / name / "data.csv", "w") as f:
            f.write("#timestamp [ns],filename\n")
            for t in img_t:
                f.write("%d,%d.png\n" % (t, t))
    for k, t in enumerate(img_t):
        idepth, rgb, _ = scene(k, rng)
        raw = np.round(1000.0 / idepth).astype(np.uint16)  # depth_scale_factor 1000
        raw[rng.random((H, W)) < 0.03] = 0
        raw[40:90, 500:600] = 0
        PIL.fromarray(rgb, mode="RGB").save(str(root / "cam0" / "data" / ("%d.png" % t)))
        PIL.fromarray(raw).save(str(root / "depth0" / "data" / ("%d.png" % t)))
    p = subprocess.run([exe, "asl", str(root / "pose"), str(root / "cam0"), str(root / "depth0"), "FLU", "200", "--dump",
                        str(tmp_path / "dump")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    rows = parse(p.stdout)
    common_checks(rows, 4)
    assert check_frames_against_oracle(str(tmp_path / "dump"), rows, 200) == 4
    # the pose that reached update(): float32 of the double-precision chain (pose sensor -> body -> camera, FLU world ->
    # optical), recomputed here with SciPy in float64 -- equal after the cast to float32 up to the last bits of the chain
    Tp = np.array([[1, 0, 0, 0.02], [0, 1, 0, -0.01], [0, 0, 1, 0.05], [0, 0, 0, 1.0]])
    Tc = np.array([[0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975],
                   [0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768],
                   [-0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949], [0, 0, 0, 1.0]])
    flu_to_rdf = np.array([[0, -1, 0], [0, 0, -1], [1, 0, 0.0]])  # x right = -y_flu, y down = -z_flu, z forward = x_flu
    for r, t_img in zip(rows, img_t):
        tp, pp, qq = min(poses, key=lambda c: abs(c[0] - t_img))  # the associated pose: nearest in time (2 ms away)
        Twp = np.eye(4); Twp[:3, :3] = R.from_quat(qq).as_matrix(); Twp[:3, 3] = pp
        Twc = Twp @ np.linalg.inv(Tp) @ Tc
        Rwc, twc = flu_to_rdf @ Twc[:3, :3], flu_to_rdf @ Twc[:3, 3]
        toks = p.stdout.splitlines()[rows.index(r)].split()
        i = toks.index("pose_t")
        t_got = np.array([float(toks[i + 1]), float(toks[i + 2]), float(toks[i + 3])])
        j = toks.index("pose_q")
        q_got = np.array([float(toks[j + 1]), float(toks[j + 2]), float(toks[j + 3]), float(toks[j + 4])])  # x y z w
        assert np.allclose(t_got, twc.astype(np.float32), atol=2e-6), (t_got, twc)
        dq = R.from_quat(q_got) * R.from_matrix(Rwc).inv()
        assert dq.magnitude() < 2e-6, dq.magnitude()
    # the colour image went through the plumb-bob rectification (sensor.yaml's distortion is not zero): without it the
    # regulariser would see the same depth features (depth is not rectified), so only the plumbing is asserted here
    assert all(r["ok"] == "1" for r in rows)
