"""Shared helpers of the test-suite (tests may use oracle/; the product may not)."""
import os
import re

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


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Fault injection lives in a library of its own (flame_ros_amd/build.py: flame_hip.cpp with -DFLAME_HIP_TEST_HOOKS=1); the
# product library has no such switch.  A test that needs it runs its scenario in a child process on that library ...
HOOKS_LIB = os.path.join(ROOT, "flame_ros_amd", "libflame_hip_hooks.so")


def hooks_env(**extra):
    """Environment of a child process that loads the hooks library instead of the product's (FLAME_HIP_LIB is the loader's
    -- flame_ros_amd/lib.py -- not the library's: the library reads no environment variable).  FLAME_HIP_HOOKS_IN_LIB=1: the
    library FLAME_HIP_LIB already names is a variant build with the hooks in it (tools/exp/build_variant.sh)."""
    lib = os.environ.get("FLAME_HIP_LIB") if os.environ.get("FLAME_HIP_HOOKS_IN_LIB") else HOOKS_LIB
    return dict(os.environ, FLAME_HIP_LIB=lib, **extra)


def with_hooks(code, **hooks):
    """... and switches the hooks on right behind its first import of the package (flame_hip_test_hook, process-wide)."""
    pre = "from flame_ros_amd import lib as _hl\n" + "".join(
        "assert _hl.load().flame_hip_test_hook(%r, %d) == 0\n" % (k.encode(), int(v)) for k, v in hooks.items())
    m = re.search(r"^from flame_ros_amd import .*\n", code, re.M)
    assert m, "no import of the package in the child's code"
    return code[:m.end()] + pre + code[m.end():]


def host_reference_opts():
    """Options a plan-only handle (device = -1: "sized as the MI355X would") needs to size like a DEVICE handle of this very
    process does right now: after a resident launch gave up -- beside another test's foreign kernels, say -- the device's lease
    sits out a back-off (uploads are sized for launches) and, when it was a one-XCD launch, drops that mode for good."""
    from flame_ros_amd.regularizer import GraphRegularizer
    with GraphRegularizer.empty(device=0) as probe:
        kw = {}
        if probe.info("persist_backoff") > 0:
            kw["persist"] = 0
        if not probe.info("one_xcd_allowed"):
            kw["one_xcd"] = 0
    return kw


def settle_lease(max_solves=5000):
    """Sit the device's back-off out (a resident launch of an EARLIER test gave up beside that test's foreign kernels): a test
    that asserts resident sizing / `persist_used` starts from a lease that is free.  Solves of a small resident-capable graph
    count the back-off down."""
    from flame_ros_amd.regularizer import GraphRegularizer, default_params
    with GraphRegularizer.empty(device=0) as probe:
        if probe.info("persist_backoff") == 0:
            return 0
    g, _ = graphgen.named("tum")
    n = 0
    with GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0) as r:
        while r.info("persist_backoff") > 0 and n < max_solves:
            r.step(default_params(), 12)
            n += 1
    return n


__all__ = ["host_reference_opts", "settle_lease", "ROOT", "HOOKS_LIB", "hooks_env", "with_hooks", "graphgen", "oracle_params", "bits", "assert_bit_equal", "make_oracle", "random_state"]
