"""Row f4: the ROS-free dataset harness include/flame_ros/dataset_streams.h against
hand-written fixtures: TUM index lines (reference src/ros_sensor_streams/
tum_rgbd_offline_stream.cc:248-300) with every input frame convention (:145-194), ASL sensor
folders (sensor.yaml + data.csv; reference src/dataset_utils/asl/dataset.h:83-103, types.h:37-120),
timestamp association (src/dataset_utils/utils.h:50-93) and the pose-sensor -> body -> camera ->
optical chain (reference src/ros_sensor_streams/asl_rgbd_offline_stream.cc:205-275).  Expected
poses are computed here independently with SciPy rotations; the last test writes a three-frame
TUM sequence WITH images (Pillow) and reads the pixels back through loadFramePixels()."""
import os
import subprocess

import numpy as np
import pytest
from scipy.spatial.transform import Rotation as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FLU_TO_RDF = R.from_quat([-0.5, 0.5, -0.5, -0.5])   # (w, x, y, z) = (-0.5, -0.5, 0.5, -0.5)
FRD_TO_RDF = R.from_matrix([[0, 1, 0], [0, 0, 1], [1, 0, 0]])
RFU_TO_RDF = R.from_matrix([[1, 0, 0], [0, 0, -1], [0, 1, 0]])


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("ds") / "dataset_streams_test")
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "dataset_streams_test.cc"), "-o", out])
    return out


def same_rotation(q_wxyz, rot, tol=1e-9):
    got = R.from_quat([q_wxyz[1], q_wxyz[2], q_wxyz[3], q_wxyz[0]])
    return (got * rot.inv()).magnitude() < tol


TUM_LINES = [
    "# pose_time tx ty tz qx qy qz qw rgb_time rgb depth_time depth",
    "1305031102.1758 1.3405 0.6266 1.6575 0.6574 0.6126 -0.2949 -0.3248 1305031102.175304 rgb/a.png 1305031102.160407 depth/a.png",
    "1305031102.2758 1.3303 0.6256 1.6464 0.6579 0.6161 -0.2932 -0.3189 1305031102.275326 rgb/b.png 1305031102.262886 depth/b.png",
    "1305031102.3758 1.3160 0.6254 1.6302 1.3 1.2 -0.6 -0.6 1305031102.375398 rgb/c.png",   # no depth, unnormalised quat
    "",
]


@pytest.mark.parametrize("frame", ["RDF", "FLU", "FRD", "RDF_IN_FLU", "RDF_IN_FRD"])
def test_tum_index_and_pose_conventions(exe, tmp_path, frame):
    idx = tmp_path / "seq" / "index.txt"
    idx.parent.mkdir()
    idx.write_text("\n".join(TUM_LINES))
    out = subprocess.run([exe, "tum", str(idx), frame], capture_output=True, text=True, check=True).stdout.splitlines()
    assert out[0].split()[:2] == ["frames", "3"]
    rows = [l.split() for l in out[1:]]
    assert [int(r[1]) for r in rows] == [0, 1, 2]
    for r, line in zip(rows, [l for l in TUM_LINES[1:] if l]):
        tok = line.split()
        assert abs(float(r[2]) - float(tok[8])) < 1e-6          # the rgb time is the frame time
        q_in = R.from_quat([float(tok[4]), float(tok[5]), float(tok[6]), float(tok[7])])  # normalises
        t_in = np.array([float(tok[1]), float(tok[2]), float(tok[3])])
        c = {"RDF": None, "FLU": FLU_TO_RDF, "FRD": FRD_TO_RDF, "RDF_IN_FLU": FLU_TO_RDF, "RDF_IN_FRD": FRD_TO_RDF}[frame]
        if c is None:
            q_exp, t_exp = q_in, t_in
        elif frame in ("FLU", "FRD"):
            q_exp, t_exp = c * q_in * c.inv(), c.apply(t_in)
        else:
            q_exp, t_exp = c * q_in, c.apply(t_in)
        assert same_rotation([float(x) for x in r[3:7]], q_exp)
        assert np.allclose([float(x) for x in r[7:10]], t_exp, atol=1e-9)
        has_depth = len(tok) >= 12
        assert int(r[10]) == int(has_depth)
        assert r[11] == str(idx.parent / tok[9])
        if has_depth:
            assert r[12] == str(idx.parent / tok[11])
        else:
            assert len(r) == 12


SENSOR_CAM = """# camera
sensor_type: camera
comment: VI-Sensor cam0 (MT9M034)
T_BS:
  cols: 4
  rows: 4
  data: [0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975,
         0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768,
        -0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949,
         0.0, 0.0, 0.0, 1.0]
rate_hz: 20
resolution: [752, 480]
camera_model: pinhole
intrinsics: [458.654, 457.296, 367.215, 248.375] #fu, fv, cu, cv
distortion_model: radial-tangential
distortion_coefficients: [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05]
"""
SENSOR_POSE = """sensor_type: pose
T_BS:
  cols: 4
  rows: 4
  data: [-0.639572038464207, -0.750418096787791, -0.166794147463654, 0.069010000000000,
         0.542717992107105, -0.287111254763649, -0.789318888936072, -0.027810000000000,
         0.544430701428539, -0.595348475601838, 0.590893733204708, -0.123950000000000,
         0.0, 0.0, 0.0, 1.0]
"""
SENSOR_DEPTH = "sensor_type: depth\ndepth_scale_factor: 1000.0\n"


def write_asl(tmp_path, with_depth):
    t0 = 1403715273262142976
    pose_t = [t0 + k * 10_000_000 for k in range(12)]                      # 100 Hz
    rgb_t = [t0 + 3_000_000 + k * 50_000_000 for k in range(3)]           # 20 Hz, 3 ms off the pose clock
    rgb_t.append(t0 + 500_000_000)                                         # no pose within 20 ms -> dropped
    rng = np.random.default_rng(0)
    poses = []
    for t in pose_t:
        q = R.from_rotvec(rng.normal(0, 0.5, 3)).as_quat()                 # x y z w
        poses.append((t, rng.normal(0, 1, 3), q * 1.7))                    # unnormalised on purpose
    for name, yaml in (("pose", SENSOR_POSE), ("cam0", SENSOR_CAM)) + ((("depth0", SENSOR_DEPTH),) if with_depth else ()):
        (tmp_path / name).mkdir()
        (tmp_path / name / "sensor.yaml").write_text(yaml)
    with open(tmp_path / "pose" / "data.csv", "w") as f:
        f.write("#timestamp [ns],p_x,p_y,p_z,q_w,q_x,q_y,q_z\n")
        for t, p, q in poses:
            f.write("%d,%.9f,%.9f,%.9f,%.9f,%.9f,%.9f,%.9f\n" % (t, p[0], p[1], p[2], q[3], q[0], q[1], q[2]))
    with open(tmp_path / "cam0" / "data.csv", "w") as f:
        f.write("#timestamp [ns],filename\n")
        for t in rgb_t:
            f.write("%d,%d.png\n" % (t, t))
    if with_depth:
        with open(tmp_path / "depth0" / "data.csv", "w") as f:
            f.write("#timestamp [ns],filename\n")
            for t in rgb_t[1:]:                                            # the first rgb frame has no depth
                f.write("%d,%d.png\n" % (t + 1_000_000, t))
    return pose_t, rgb_t, poses


@pytest.mark.parametrize("world,with_depth", [("RDF", False), ("FLU", True), ("FRD", False), ("RFU", True)])
def test_asl_dataset_association_and_pose_chain(exe, tmp_path, world, with_depth):
    pose_t, rgb_t, poses = write_asl(tmp_path, with_depth)
    depth_arg = str(tmp_path / "depth0") if with_depth else "-"
    out = subprocess.run([exe, "asl", str(tmp_path / "pose") + "/", str(tmp_path / "cam0"), depth_arg, world],
                         capture_output=True, text=True, check=True).stdout.splitlines()
    hdr = out[0].split()
    expect_rgb = [0, 1, 2] if not with_depth else [1, 2]                   # 4th rgb frame has no pose nearby
    assert int(hdr[1]) == len(expect_rgb) and (int(hdr[3]), int(hdr[5])) == (752, 480)
    assert np.allclose([float(x) for x in hdr[7:11]], [458.654, 457.296, 367.215, 248.375])
    assert np.allclose([float(x) for x in hdr[12:17]], [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0])
    assert float(hdr[18]) == (1000.0 if with_depth else 0.0)
    Tp = np.array([float(x) for x in SENSOR_POSE.split("[")[1].split("]")[0].replace("\n", " ").split(",")]).reshape(4, 4)
    Tc = np.array([float(x) for x in SENSOR_CAM.split("data: [")[1].split("]")[0].replace("\n", " ").split(",")]).reshape(4, 4)
    q_pb, t_pb = R.from_matrix(Tp[:3, :3]), Tp[:3, 3]
    q_cb, t_cb = R.from_matrix(Tc[:3, :3]), Tc[:3, 3]
    conv = {"RDF": None, "FLU": FLU_TO_RDF, "FRD": FRD_TO_RDF, "RFU": RFU_TO_RDF}[world]
    rows = [l.split() for l in out[1:]]
    assert len(rows) == len(expect_rgb)
    for k, (r, ri) in enumerate(zip(rows, expect_rgb)):
        assert int(r[1]) == k and abs(float(r[2]) - rgb_t[ri] * 1e-9) < 1e-6
        pi = int(np.argmin(np.abs(np.array(pose_t, np.int64) - rgb_t[ri])))  # closest pose sample
        _, p, q = poses[pi]
        q_pw = R.from_quat(q)                                               # normalises
        q_bp, t_bp = q_pb.inv(), -(q_pb.inv().apply(t_pb))
        q_bw, t_bw = q_pw * q_bp, q_pw.apply(t_bp) + p
        q_cw, t_cw = q_bw * q_cb, q_bw.apply(t_cb) + t_bw
        if conv is not None:
            q_cw, t_cw = conv * q_cw, conv.apply(t_cw)
        # the camera's T_BS is written with 12 digits: orthonormal to ~1e-10 only
        assert same_rotation([float(x) for x in r[3:7]], q_cw, 1e-7)
        assert np.allclose([float(x) for x in r[7:10]], t_cw, atol=1e-7)
        assert r[10] == str(tmp_path / "cam0" / "data" / ("%d.png" % rgb_t[ri]))
        if with_depth:
            assert r[11] == str(tmp_path / "depth0" / "data" / ("%d.png" % rgb_t[ri]))


def test_tum_sequence_with_pixels(exe, tmp_path):
    """Index -> file names -> decoded gray image + depth in metres, as flame_offline_tum hands them
    to update() (reference src/flame_offline_tum.cc:565-594; decode / rectify / scale:
    src/ros_sensor_streams/tum_rgbd_offline_stream.cc:196-209)."""
    PIL = pytest.importorskip("PIL.Image")
    seq = tmp_path / "seq"
    (seq / "rgb").mkdir(parents=True)
    (seq / "depth").mkdir()
    (seq / "index.txt").write_text("\n".join(TUM_LINES))
    rng = np.random.default_rng(0)
    imgs, depths = {}, {}
    for name in "abc":
        rgb = (rng.random((48, 64, 3)) * 255).astype(np.uint8)
        rgb[10:30, 20:40] = 200
        PIL.fromarray(rgb, mode="RGB").save(str(seq / "rgb" / (name + ".png")))
        imgs[name] = rgb
        if name != "c":  # the third index line has no depth image
            d = (rng.random((48, 64)) * 20000).astype(np.uint16)
            PIL.fromarray(d).save(str(seq / "depth" / (name + ".png")))
            depths[name] = d
    out = str(tmp_path / "pix.bin")
    subprocess.run([exe, "tumpix", str(seq / "index.txt"), "RDF", out], check=True)
    raw, off = open(out, "rb").read(), 0
    for name in "abc":
        w, h, has_d = np.frombuffer(raw, np.int32, 3, off)
        off += 12
        assert (w, h) == (64, 48) and bool(has_d) == (name in depths)
        gray = np.frombuffer(raw, np.uint8, w * h, off).reshape(h, w)
        off += w * h
        r, g, b = (imgs[name][..., k].astype(np.int64) for k in range(3))
        assert np.array_equal(gray, ((4899 * r + 9617 * g + 1868 * b + 8192) >> 14).astype(np.uint8))
        if has_d:
            d = np.frombuffer(raw, np.float32, w * h, off).reshape(h, w)
            off += 4 * w * h
            assert np.array_equal(d, depths[name].astype(np.float32) / np.float32(5000))
    assert off == len(raw)
    # a missing image is an error, not a crash
    os.remove(str(seq / "rgb" / "b.png"))
    assert subprocess.run([exe, "tumpix", str(seq / "index.txt"), "RDF", out], capture_output=True).returncode == 5
