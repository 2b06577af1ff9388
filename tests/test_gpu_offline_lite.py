"""-m gpu: BASELINE config 1's plumbing end to end without ROS -- a TUM-format sequence on disk
(index file, 8-bit RGB PNGs, 16-bit depth PNGs of a known two-plane scene with depth holes) through
tools/flame_offline_lite.cc: dataset index -> PNG decode -> flame::Flame::update() with a FrontEnd
(grid features with idepth from the depth image: the stand-in for upstream's feature pipeline) ->
idepth mesh, stats.  The regularised idepths must stay close to the scene's true idepths and every
frame must come back ok.  Reference flow: src/flame_offline_tum.cc:565-707."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tum_sequence_through_the_facade(gpu, tmp_path):
    PIL = pytest.importorskip("PIL.Image")
    exe = str(tmp_path / "flame_offline_lite")
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "flame_offline_lite.cc"), "-o", exe,
                           "-L" + os.path.join(ROOT, "flame_ros_amd"), "-lflame_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "flame_ros_amd"), "-pthread"])
    seq = tmp_path / "seq"
    (seq / "rgb").mkdir(parents=True)
    (seq / "depth").mkdir()
    W, H, fx, fy, cx, cy = 640, 480, 525.0, 525.0, 319.5, 239.5
    rng = np.random.default_rng(0)
    vv, uu = np.mgrid[0:H, 0:W]
    lines = ["# synthetic sequence"]
    for k in range(4):
        # two fronto-parallel-ish planes: idepth affine in (u, v), a step at the image centre
        idepth = 0.45 + 0.0004 * uu - 0.0002 * vv + 0.25 * (uu > W // 2 + 10 * k)
        depth = 1.0 / idepth
        raw = np.round(depth * 5000).astype(np.uint16)
        raw[rng.random((H, W)) < 0.03] = 0           # holes: no measurement (no feature there)
        raw[40:90, 500:600] = 0
        rgb = np.clip((idepth[..., None] * 160 + rng.normal(0, 6, (H, W, 3))), 0, 255).astype(np.uint8)
        PIL.fromarray(rgb, mode="RGB").save(str(seq / "rgb" / ("%d.png" % k)))
        PIL.fromarray(raw).save(str(seq / "depth" / ("%d.png" % k)))
        t = 1305031102.175304 + 0.033 * k
        lines.append("%.6f 1.34 0.62 1.65 0.6574 0.6126 -0.2949 -0.3248 %.6f rgb/%d.png %.6f depth/%d.png" % (t, t, k, t, k))
    (seq / "index.txt").write_text("\n".join(lines) + "\n")
    p = subprocess.run([exe, str(seq / "index.txt"), "RDF", str(fx), str(fy), str(cx), str(cy), "200"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    rows = [dict(zip(l.split()[0::2], l.split()[1::2])) for l in p.stdout.splitlines() if l.startswith("frame")]
    assert len(rows) == 4
    for r in rows:
        assert r["ok"] == "1" and r["hip_error"] == "0"
        assert 1000 <= int(r["vtx"]) <= 1200 and int(r["feats"]) == int(r["vtx"])  # 40 x 30 cells minus the holes
        assert int(r["tris"]) > 1800 and int(r["edges"]) > 2800
        assert 0.6 < float(r["coverage"]) <= 1.0
        # the regulariser smooths noise-free plane data only at the step: close to the truth
        assert float(r["rms_vs_truth"]) < 0.03, r
        if r is not rows[0]:  # (the first update creates the GPU context, streams and buffers)
            assert float(r["update_ms"]) < 5.0
    # a missing image file is reported, not a crash
    os.remove(str(seq / "rgb" / "2.png"))
    p = subprocess.run([exe, str(seq / "index.txt"), "RDF", str(fx), str(fy), str(cx), str(cy)], capture_output=True, text=True)
    assert p.returncode == 4 and "cannot read" in p.stderr
