"""Generates the golden fixtures under tests/golden/ (run once in the build container).

There are NO golden vectors in the reference (it has no tests: reference CMakeLists.txt:271-281),
and the solver source is not under /root/reference, so these fixtures are produced by the build's
own oracle (oracle/nltgv2_oracle.c, PARITY UNPINNED) after it has been pinned by the analytic
known-answer tests and the float64 cross-check in tests/test_oracle_kat.py.  They freeze the
oracle: any later change of its arithmetic shows up as a golden mismatch.

  g5k.npz        BASELINE config 2 graph (5000 vertices, seed 0): pos, edges, tris, alpha, z and
                 the oracle state x after 1 / 10 / 200 iterations (+ w1, w2, q after 200)
  g12.npz        12-vertex hand-checkable graph, state after 1 / 5 iterations
  two_vertex.npz K4: one edge, closed-form first iteration
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flame_ros_amd import graphgen  # noqa: E402
from oracle import COracle  # noqa: E402
from oracle.cbind import default_params  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    p = default_params()
    g = graphgen.synthetic(5000, seed=0)
    o = COracle(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt)
    out = dict(pos=g.pos, edges=g.edges, tris=g.tris, alpha=g.alpha, beta=g.beta, z=g.z, wgt=g.wgt)
    done = 0
    for n in (1, 10, 200):
        o.solve(p, n - done)
        done = n
        out["x_after_%d" % n] = o.x.copy()
    out.update(w1_after_200=o.w1.copy(), w2_after_200=o.w2.copy(), q_after_200=o.q.copy(),
               costs_after_200=np.array(o.costs(p)))
    np.savez_compressed(os.path.join(HERE, "g5k.npz"), **out)

    rng = np.random.default_rng(12)
    pos = np.array([[x * 40.0 + rng.uniform(-8, 8), y * 40.0 + rng.uniform(-8, 8)]
                    for y in range(3) for x in range(4)], np.float32)
    g12 = graphgen.from_points(pos, 160, 120, np.random.Generator(np.random.PCG64(12)))
    o = COracle(g12.pos, g12.edges, g12.alpha, g12.beta, g12.z, g12.wgt)
    o.solve(p, 1)
    x1 = o.x.copy()
    o.solve(p, 4)
    np.savez_compressed(os.path.join(HERE, "g12.npz"), pos=g12.pos, edges=g12.edges, tris=g12.tris,
                        alpha=g12.alpha, beta=g12.beta, z=g12.z, wgt=g12.wgt, x_after_1=x1,
                        x_after_5=o.x.copy(), w1_after_5=o.w1.copy(), q_after_5=o.q.copy())

    # K4 two-vertex closed form, computed here in float64 from the formulas (not by the oracle)
    z0, z1, a, b, dx, dy = 0.8, 0.3, 0.05, 0.05, -20.0, 0.0
    sig, tau, lam, th = p.step_q, p.step_x, p.data_factor, p.theta
    q1 = np.clip(sig * a * (z0 - z1), -1, 1)
    x0, x1v = z0 - tau * a * q1, z1 + tau * a * q1
    w1_0 = -tau * (-a * dx * q1)
    t = tau * lam

    def prox(x, z):
        r = x - z
        return x - t if r > t else (x + t if r < -t else z)
    x0p, x1p = prox(x0, z0), prox(x1v, z1)
    np.savez(os.path.join(HERE, "two_vertex.npz"), z=np.array([z0, z1]), alpha=a, beta=b,
             pos=np.array([[100.0, 50.0], [100.0 - dx, 50.0 - dy]]), q1=q1, x=np.array([x0p, x1p]),
             w1_0=w1_0, xb=np.array([x0p + th * (x0p - z0), x1p + th * (x1p - z1)]))
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
