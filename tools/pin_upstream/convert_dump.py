"""FLDUMP1 (tools/pin_upstream/fldump.h) -> the npz layout tests/test_upstream_pin.py consumes.

  python tools/pin_upstream/convert_dump.py dump.fldump tests/golden/upstream_<tag>.npz [key=value ...]

key=value pairs become the dump's switches (all optional): d_sign=-1, source="flame@<commit>".
Arrays: pos[V,2] edges[E,2] alpha[E] beta[E] z[V] wgt[V] x0[V] params[6] = (data_factor, step_x,
step_q, theta, x_min, x_max), iters[K], and per recorded iteration count n: x_after_n, w1_after_n,
w2_after_n [V], q_after_n [E,3].  A FRAME dump (dump_upstream_frame.cc: mesh_*, raw_*, idepthmap_filtered ...) is
passed through under the same rules (tests/golden/upstream_frame_<tag>.npz)."""
import struct
import sys

import numpy as np


def read_dump(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"FLDUMP1\n", "not an FLDUMP1 file"
    off, out = 8, {}
    while off < len(raw):
        end = raw.index(b"\n", off)
        name = raw[off:end].decode()
        dtype = chr(raw[end + 1])
        (ndim,) = struct.unpack_from("<I", raw, end + 2)
        shape = struct.unpack_from("<%dI" % ndim, raw, end + 6)
        off = end + 6 + 4 * ndim
        n = int(np.prod(shape))
        out[name] = np.frombuffer(raw, np.float32 if dtype == "f" else np.int32, n, off).reshape(shape).copy()
        off += 4 * n
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    d = read_dump(src)
    for kv in sys.argv[3:]:
        k, v = kv.split("=", 1)
        d[k] = np.array(int(v)) if v.lstrip("-").isdigit() else np.array(v)
    if "mesh_pos" in d:  # a FRAME dump (dump_upstream_frame.cc): what leaves flame::Flame after an update
        need = ("image_size", "K", "rparams", "tri_filter", "tri_filter_on", "sync", "raw_pos", "raw_mu", "raw_var", "mesh_pos",
                "mesh_idepth", "mesh_normals", "mesh_tris", "mesh_tri_valid", "mesh_edges", "idepthmap_filtered")
        missing = [k for k in need if k not in d]
        assert not missing, "frame dump lacks %s" % missing
        np.savez_compressed(dst, **d)
        print("%s: frame dump, %d raw features, mesh V=%d T=%d E=%d, map %s" % (
            dst, len(d["raw_mu"]), len(d["mesh_idepth"]), len(d["mesh_tris"]), len(d["mesh_edges"]), d["idepthmap_filtered"].shape))
        return
    need = ("pos", "edges", "alpha", "beta", "z", "wgt", "x0", "params", "iters")
    missing = [k for k in need if k not in d]
    assert not missing, "dump lacks %s" % missing
    for n in d["iters"]:
        assert "x_after_%d" % n in d, "dump lacks the state after %d iterations" % n
    np.savez_compressed(dst, **d)
    print("%s: V=%d E=%d iterations %s" % (dst, len(d["z"]), len(d["alpha"]), d["iters"].tolist()))


if __name__ == "__main__":
    main()
