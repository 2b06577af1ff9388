"""r06 dev aid: the stamps of tools/exp/iter_prof_build.py's variant library, read on the GPU box.
  FLAME_HIP_LIB=flame_ros_amd/libflame_hip_itp.so python tools/exp/iter_prof.py [names...]
For each graph: three tiles (the one with the median local edge count, the largest, the smallest), one round in the middle of
the solve; per iteration of the round the wave-by-wave split in shader cycles."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flame_ros_amd import graphgen, lib as _l  # noqa: E402
from flame_ros_amd.regularizer import GraphRegularizer, default_params  # noqa: E402

L = _l.load()
fn = L.flame_hip_exp_itp
fn.restype, fn.argtypes = C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int]
p = default_params()
NAMES = ["0 top->gathers landed", "1 dual ascent + slot stores drained", "2 wait at barrier 1", "3 slot reads + ordered sums",
         "4 prox, extrapolate, bar store drained", "5 wait at barrier 2"]
for name in (sys.argv[1:] or ["tum", "5k", "euroc", "50k"]):
    g, it = graphgen.named(name)
    with GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0) as r:
        r.step(p, it)
        r.step(p, it)
        nt, depth = r.info("num_tiles"), r.info("tile_depth")
        t = r.plan_array("tiles", np.int32).reshape(-1, 47)
        order = np.argsort(t[:, 5])
        print("== %s: V %d, %d tiles, depth %d, threads %d, ept %d, resident %d" % (name, g.V, nt, depth, r.info("tile_threads"), r.info("tile_ept"), r.info("persist_used")))
        for label, tile in (("median", int(order[nt // 2])), ("largest", int(order[-1])), ("smallest", int(order[0]))):
            assert fn(tile, 3, None, 0) == 0
            r.step(p, it)
            buf = np.zeros(16 * 8 * 8, np.uint64)
            assert fn(0, 0, buf.ctypes.data_as(C.c_void_p), 1) == 0
            s = buf.reshape(16, 8, 8).astype(np.int64)
            d = t[tile]
            print("-- %s tile %d: n_own %d n_ext %d n_upd %d e_own %d e_loc %d nslots %d; ring_end %s level_end %s" % (
                label, tile, d[1], d[2], d[6], d[4], d[5], d[12], list(d[13:13 + depth + 1]), list(d[30:30 + depth + 1])))
            waves = [w for w in range(16) if s[w, 0, 0] != 0]
            t0 = min(s[w, 0, 0] for w in waves)
            for i in range(min(depth, 8)):
                # the iteration as the workgroup sees it: from the first wave's top to the last wave behind barrier 2
                top = min(s[w, i, 0] for w in waves)
                end = max(s[w, i, 6] for w in waves)
                b1 = max(s[w, i, 3] for w in waves)
                print("   iteration %d: %5d cycles (phase D + barrier %5d, phase P + barrier %5d)  [starts at %d]" % (
                    i + 1, end - top, b1 - top, end - b1, top - t0))
                for w in waves:
                    x = s[w, i]
                    has_d, has_p = x[1] != 0, x[4] != 0
                    seg = []
                    seg.append("gather %4d" % (x[1] - x[0]) if has_d else "gather    -")
                    seg.append("dual+stores %4d" % (x[2] - (x[1] if has_d else x[0])))
                    seg.append("bar1 %4d" % (x[3] - x[2]))
                    seg.append("slots %4d" % (x[4] - x[3]) if has_p else "slots    -")
                    if has_p and x[7]: seg.append("(reads landed %4d, chain %4d)" % (x[7] - x[3], x[4] - x[7]))
                    seg.append("prox+store %4d" % (x[5] - x[4]) if has_p else "prox+store    -")
                    seg.append("bar2 %4d" % (x[6] - (x[5] if has_p else x[3])))
                    print("      wave %2d: %s" % (w, " | ".join(seg)))
