"""Row f4 (pixel side of the ROS-free dataset harness): include/flame_ros/image_io.h against Pillow
(decode) and a NumPy statement of the same float32 rule (rectification).  CPU only.  What it stands
in for: cv::imread / cv::undistort / the depth scaling of flame_ros' offline streams (reference
src/ros_sensor_streams/tum_rgbd_offline_stream.cc:196-209, asl_rgbd_offline_stream.cc:282-308)."""
import os
import subprocess
import zlib

import numpy as np
import pytest

PIL = pytest.importorskip("PIL.Image")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("imgio") / "image_io_test")
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "image_io_test.cc"), "-o", out])
    return out


def run(exe, mode, src, out, *args):
    p = subprocess.run([exe, mode, src, out] + [repr(float(a)) for a in args], capture_output=True, text=True)
    assert p.returncode == 0, (p.returncode, p.stderr)
    raw = open(out, "rb").read()
    w, h, c, bd = np.frombuffer(raw[:16], np.int32)
    return np.frombuffer(raw[16:], np.uint8 if bd == 8 else np.uint16).reshape(h, w, c)


def scene(h, w, c, dtype, seed):
    """A smooth image with texture and noise (so the PNG encoder picks several row filters)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    top = np.iinfo(dtype).max
    chans = []
    for k in range(c):
        a = 0.5 + 0.3 * np.sin(xx / (7.0 + k)) * np.cos(yy / (11.0 - k)) + 0.1 * rng.random((h, w))
        a[h // 3:h // 2, w // 4:w // 2] = 0.9  # a flat block (long matches: distance codes)
        chans.append(np.clip(a * top, 0, top).astype(dtype))
    return np.stack(chans, -1)


@pytest.mark.parametrize("mode,c,dtype", [("L", 1, np.uint8), ("RGB", 3, np.uint8), ("RGBA", 4, np.uint8),
                                          ("I;16", 1, np.uint16), ("LA", 2, np.uint8)])
@pytest.mark.parametrize("level", [0, 1, 9])  # stored / fast (fixed or dynamic) / best (dynamic Huffman)
def test_png_decode_matches_pillow(exe, tmp_path, mode, c, dtype, level):
    img = scene(97, 131, c, dtype, seed=c)
    path = str(tmp_path / "a.png")
    PIL.fromarray(img[..., 0] if c == 1 else img, mode=mode).save(path, compress_level=level)
    got = run(exe, "decode", path, str(tmp_path / "o.bin"))
    want = np.asarray(PIL.open(path))
    want = want.reshape(97, 131, c).astype(dtype)
    assert got.dtype == want.dtype and np.array_equal(got, want)
    assert np.array_equal(got, img)


def test_png_row_filters_and_fixed_huffman(exe, tmp_path):
    """A hand-assembled PNG that uses each of the five row filters once per five rows and a
    fixed-Huffman deflate stream (zlib strategy Z_FIXED), independent of what Pillow's encoder picks."""
    import struct
    h, w = 25, 40
    img = scene(h, w, 3, np.uint8, seed=9)
    bpp, stride = 3, 3 * w
    raw = bytearray()
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        cur = img[y].reshape(-1).astype(np.int32)
        ft = y % 5
        a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        cc = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if ft == 0: pred = np.zeros(stride, np.int32)
        elif ft == 1: pred = a
        elif ft == 2: pred = prev
        elif ft == 3: pred = (a + prev) >> 1
        else:
            p = a + prev - cc
            pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - cc)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, cc))
        raw.append(ft)
        raw += bytes(((cur - pred) & 255).astype(np.uint8))
        prev = cur
    co = zlib.compressobj(9, zlib.DEFLATED, 15, 9, zlib.Z_FIXED)
    z = co.compress(bytes(raw)) + co.flush()

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xffffffff)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + \
        chunk(b"IDAT", z[:len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b"")
    path = str(tmp_path / "f.png")
    open(path, "wb").write(png)
    assert np.array_equal(np.asarray(PIL.open(path)), img)  # the file is a valid PNG
    assert np.array_equal(run(exe, "decode", path, str(tmp_path / "o.bin")), img)


def test_pnm_and_gray_conversion(exe, tmp_path):
    img = scene(60, 80, 3, np.uint8, seed=4)
    ppm = str(tmp_path / "a.ppm")
    open(ppm, "wb").write(b"P6\n# a comment\n80 60\n255\n" + img.tobytes())
    assert np.array_equal(run(exe, "decode", ppm, str(tmp_path / "o.bin")), img)
    pgm16 = str(tmp_path / "d.pgm")
    d16 = scene(60, 80, 1, np.uint16, seed=5)
    open(pgm16, "wb").write(b"P5 80 60 65535\n" + d16.astype(">u2").tobytes())
    assert np.array_equal(run(exe, "decode", pgm16, str(tmp_path / "o.bin")), d16)
    # BGR2GRAY with OpenCV's fixed-point weights
    g = run(exe, "gray", ppm, str(tmp_path / "g.bin"))[..., 0]
    r, gg, b = (img[..., k].astype(np.int64) for k in range(3))
    assert np.array_equal(g, ((4899 * r + 9617 * gg + 1868 * b + 8192) >> 14).astype(np.uint8))
    # depth scaling (TUM: 5000 units per metre)
    out = str(tmp_path / "m.bin")
    assert subprocess.run([exe, "depth", pgm16, out, "5000"]).returncode == 0
    m = np.fromfile(out, np.float32).reshape(60, 80)
    assert np.array_equal(m, d16[..., 0].astype(np.float32) / np.float32(5000))


def undistort_np(src, cam):
    """The rule of image_io.h undistort(), float32 step for step."""
    f = np.float32
    fx, fy, cx, cy, k1, k2, p1, p2, k3 = (f(v) for v in cam)
    h, w, c = src.shape
    v, u = np.mgrid[0:h, 0:w].astype(np.float32)
    x, y = (u - cx) / fx, (v - cy) / fy
    r2 = x * x + y * y
    radial = f(1) + r2 * (k1 + r2 * (k2 + r2 * k3))
    xd = x * radial + f(2) * p1 * x * y + p2 * (r2 + f(2) * x * x)
    yd = y * radial + p1 * (r2 + f(2) * y * y) + f(2) * p2 * x * y
    su, sv = fx * xd + cx, fy * yd + cy
    x0, y0 = np.floor(su), np.floor(sv)
    ax, ay = su - x0, sv - y0
    x0, y0 = x0.astype(np.int64), y0.astype(np.int64)

    def at(xx, yy):
        ok = (xx >= 0) & (yy >= 0) & (xx < w) & (yy < h)
        return np.where(ok[..., None], src[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.float32), f(0))
    ax, ay = ax[..., None], ay[..., None]
    top = at(x0, y0) + ax * (at(x0 + 1, y0) - at(x0, y0))
    bot = at(x0, y0 + 1) + ax * (at(x0 + 1, y0 + 1) - at(x0, y0 + 1))
    return (top + ay * (bot - top) + f(0.5)).astype(src.dtype)


@pytest.mark.parametrize("mode,c,dtype", [("L", 1, np.uint8), ("RGB", 3, np.uint8), ("I;16", 1, np.uint16)])
def test_rectification(exe, tmp_path, mode, c, dtype):
    img = scene(120, 160, c, dtype, seed=7)
    path = str(tmp_path / "a.png")
    PIL.fromarray(img[..., 0] if c == 1 else img, mode=mode).save(path)
    euroc = (458.654 / 4, 457.296 / 4, 367.215 / 4.7, 248.375 / 4, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0)
    got = run(exe, "rectify", path, str(tmp_path / "o.bin"), *euroc)
    want = undistort_np(img, euroc)
    diff = np.abs(got.astype(np.int64) - want.astype(np.int64))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, (diff.max(), (diff > 0).mean())  # (x87-free float32: equal up to rounding ties)
    assert (got != img).mean() > 0.5  # the distortion really moved pixels
    # no distortion => the identity
    ident = run(exe, "rectify", path, str(tmp_path / "i.bin"), 100.0, 100.0, 80.0, 60.0, 0, 0, 0, 0, 0)
    assert np.array_equal(ident, img)


def test_corrupt_files_are_rejected(exe, tmp_path):
    path = str(tmp_path / "a.png")
    PIL.fromarray(scene(20, 30, 1, np.uint8, 1)[..., 0], mode="L").save(path)
    good = open(path, "rb").read()
    for name, data in (("trunc.png", good[:len(good) // 2]), ("magic.png", b"XX" + good[2:]),
                       ("zlib.png", good[:50] + bytes([good[50] ^ 0xff]) + good[51:])):
        bad = str(tmp_path / name)
        open(bad, "wb").write(data)
        p = subprocess.run([exe, "decode", bad, str(tmp_path / "o.bin")], capture_output=True)
        assert p.returncode == 3, name


def test_crafted_headers_are_rejected(exe, tmp_path):
    """IHDR / PNM dimensions are untrusted.  A PNG whose IHDR claims 2^31 - 1 x 2^31 - 1 (sizes that wrap
    in size_t arithmetic), one whose IHDR claims a tiny image in front of a zlib stream that inflates to megabytes,
    and PNM headers with absurd dimensions are all refused (exit 3), none is followed into an allocation or a copy."""
    import struct, zlib

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xffffffff)

    def png(w, h, payload):
        return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(payload)) + chunk(b"IEND", b""))
    cases = {"huge.png": png(0x7fffffff, 0x7fffffff, b"\0" * 64),
             "wide.png": png(1 << 20, 4, b"\0" * 64),
             "bomb.png": png(4, 4, b"\0" * (8 << 20)),
             "huge.pgm": b"P5\n2147483647 2147483647\n255\n" + b"\0" * 16,
             "wide.pgm": b"P5\n99999999999999999999 3\n255\n" + b"\0" * 16,
             "zero.pgm": b"P5\n0 3\n255\n"}
    for name, data in cases.items():
        bad = str(tmp_path / name)
        open(bad, "wb").write(data)
        p = subprocess.run([exe, "decode", bad, str(tmp_path / "o.bin")], capture_output=True, timeout=60)
        assert p.returncode == 3, (name, p.returncode)


@pytest.mark.parametrize("seed", range(6))
def test_inflate_against_zlib_streams(exe, tmp_path, seed):
    """The header-only inflate against streams zlib produced with every strategy / level / window size
    that changes the block structure (stored, fixed Huffman, dynamic Huffman, RLE, long distances),
    incl. the empty input and inputs that span many deflate blocks."""
    rng = np.random.default_rng(seed)
    kinds = [b"", bytes(rng.integers(0, 256, 70000, dtype=np.uint8)),                      # incompressible
             bytes(rng.integers(0, 4, 200000, dtype=np.uint8)),                            # low entropy
             (b"flame_ros " * 5000) + bytes(rng.integers(0, 256, 999, dtype=np.uint8)),   # long matches
             bytes(np.repeat(rng.integers(0, 256, 300, dtype=np.uint8), rng.integers(1, 400, 300)))]  # runs
    data = kinds[seed % len(kinds)] if seed < len(kinds) else b"".join(kinds)
    for level, strategy, wbits in ((0, zlib.Z_DEFAULT_STRATEGY, 15), (1, zlib.Z_DEFAULT_STRATEGY, 15),
                                   (9, zlib.Z_DEFAULT_STRATEGY, 15), (6, zlib.Z_FIXED, 15), (6, zlib.Z_RLE, 15),
                                   (6, zlib.Z_HUFFMAN_ONLY, 15), (9, zlib.Z_FILTERED, 9)):
        co = zlib.compressobj(level, zlib.DEFLATED, wbits, 9, strategy)
        z = co.compress(data) + co.flush()
        src, out = str(tmp_path / "z.bin"), str(tmp_path / "o.bin")
        open(src, "wb").write(z)
        p = subprocess.run([exe, "inflate", src, out], capture_output=True)
        assert p.returncode == 0, (level, strategy, wbits, len(data))
        assert open(out, "rb").read() == data, (level, strategy, wbits)
    # a truncated stream is rejected
    open(src, "wb").write(z[:max(3, len(z) // 2)])
    assert subprocess.run([exe, "inflate", src, out], capture_output=True).returncode in (6,) or len(data) == 0
