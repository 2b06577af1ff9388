"""Independent float64 NumPy restatement of the NLTGV2-L1 primal-dual iteration.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Written from the paper-level formulas in SURVEY.md
section 8a rows a2-a6 in vectorised form (scatter by np.add.at), deliberately NOT sharing code or
operation order with nltgv2_oracle.c: agreement of the two (K9: <= 1e-5 RMS after 200 iterations)
is what pins the C restatement in the absence of reference golden vectors (PARITY UNPINNED).
"""
import numpy as np


class NpSolver:
    def __init__(self, pos, edges, alpha, beta, z, wgt, x0=None, dtype=np.float64):
        f = lambda a: np.asarray(a, dtype=dtype)  # noqa: E731
        self.pos = f(pos).reshape(-1, 2)
        self.edges = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
        self.alpha, self.beta, self.z, self.wgt = f(alpha), f(beta), f(z), f(wgt)
        self.V, self.E = len(self.z), len(self.alpha)
        self.i, self.j = self.edges[:, 0], self.edges[:, 1]
        self.d = self.pos[self.i] - self.pos[self.j]
        self.x = (self.z if x0 is None else f(x0)).copy()
        self.w = np.zeros((self.V, 2), dtype)
        self.xb, self.wb = self.x.copy(), self.w.copy()
        self.q = np.zeros((self.E, 3), dtype)

    def K(self, x, w):
        i, j = self.i, self.j
        k1 = self.alpha * (x[i] - x[j] - (w[i] * self.d).sum(1))
        k23 = self.beta[:, None] * (w[i] - w[j])
        return np.column_stack([k1, k23])

    def KT(self, q):
        i, j = self.i, self.j
        kx = np.zeros(self.V, q.dtype)
        kw = np.zeros((self.V, 2), q.dtype)
        aq = self.alpha * q[:, 0]
        bq = self.beta[:, None] * q[:, 1:]
        np.add.at(kx, i, aq)
        np.add.at(kx, j, -aq)
        np.add.at(kw, i, -aq[:, None] * self.d + bq)
        np.add.at(kw, j, -bq)
        return kx, kw

    def step(self, lam, tau, sigma, theta, x_min=0.0, x_max=10.0):
        v = self.q + sigma * self.K(self.xb, self.wb)
        self.q = v / np.maximum(1.0, np.abs(v))
        xp, wp = self.x, self.w
        kx, kw = self.KT(self.q)
        x = xp - tau * kx
        w = wp - tau * kw
        t = tau * lam * self.wgt
        r = x - self.z
        x = np.where(r > t, x - t, np.where(r < -t, x + t, self.z))
        x = np.clip(x, x_min, x_max)
        self.x, self.w = x, w
        self.xb = x + theta * (x - xp)
        self.wb = w + theta * (w - wp)

    def solve(self, n, lam=0.15, tau=1e-3, sigma=125.0, theta=0.25, x_min=0.0, x_max=10.0):
        for _ in range(n):
            self.step(lam, tau, sigma, theta, x_min, x_max)
        return self.x

    def costs(self, lam=0.15):
        k = self.K(self.x, self.w)
        smooth = np.abs(k).sum()
        data = (lam * self.wgt * np.abs(self.x - self.z)).sum()
        return float(smooth), float(data)
