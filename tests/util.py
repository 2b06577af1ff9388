"""Shared helpers of the test-suite (tests may use oracle/; the product may not)."""
import numpy as np

from flame_ros_amd import graphgen
from oracle import COracle
from oracle.cbind import default_params as oracle_params


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(got, want, what):
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    assert got.shape == want.shape, what
    bad = bits(got) != bits(want)
    if bad.any():
        idx = np.argwhere(bad)[:5]
        raise AssertionError("%s: %d / %d words differ (max abs %.3e), first at %s" % (
            what, int(bad.sum()), bad.size, float(np.abs(got - want).max()), idx.tolist()))


def make_oracle(g, x0=None):
    return COracle(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, x0=x0)


def random_state(g, seed):
    """A generic non-trivial solver state (exercises every term, incl. saturated duals)."""
    rng = np.random.default_rng(seed)
    V, E = g.V, g.E
    st = dict(x=g.z + rng.normal(0, 0.05, V), w1=rng.normal(0, 1e-3, V), w2=rng.normal(0, 1e-3, V))
    st["xb"] = st["x"] + rng.normal(0, 0.01, V)
    st["w1b"] = st["w1"] + rng.normal(0, 1e-4, V)
    st["w2b"] = st["w2"] + rng.normal(0, 1e-4, V)
    st["q"] = np.clip(rng.normal(0, 0.8, (E, 3)), -1, 1)
    return {k: v.astype(np.float32) for k, v in st.items()}


__all__ = ["graphgen", "oracle_params", "bits", "assert_bit_equal", "make_oracle", "random_state"]
