"""r06: resident graphs of up to 32 tiles keep their tiles on ONE XCD and hand over through ordinary memory -- that XCD's L2 --
instead of uncached memory (flame_hip option "one_xcd", default 1; csrc/flame_hip.cpp, kernels.hip PersistArgs::one_xcd).
Where a workgroup lands is the dispatcher's habit, not a guarantee: the tags + bounded polls keep the result right either
way (tiles that cannot see each other time out, the solve is repeated by launches, the mode is dropped for the process)."""
import subprocess
import sys

import numpy as np
import pytest

from flame_ros_amd.regularizer import GraphRegularizer, default_params, default_sync_params
from tests.util import ROOT, assert_bit_equal, graphgen, hooks_env, make_oracle, oracle_params, with_hooks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("V", [700, 1000, 1200, 1280])
def test_one_xcd_graphs_match_the_oracle(gpu, V):
    g = graphgen.dataset_shaped(640, 480, 16) if V == 1200 else graphgen.synthetic(V, seed=V)
    p = default_params()
    o = make_oracle(g)
    with GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0) as r:
        assert 2 <= r.info("num_tiles") <= 32, r.info("num_tiles")
        for n in (60, 45, 200):  # a plan's first solve (poll lists in local order), then the address-sorted lists
            r.step(p, n)
            o.solve(oracle_params(), n)
            assert r.info("persist_used") == 1 and r.info("one_xcd_used") == 1
            x, w1, w2, q = r.download()
            for a, b, nm in ((x, o.x, "x"), (w1, o.w1, "w1"), (w2, o.w2, "w2"), (q, o.q, "q")):
                assert_bit_equal(a, b, "%s after %d more iterations" % (nm, n))
        assert r.info("persist_recovered") == 0
    # the option switched off: the placement-independent tiles (uncached hand-offs), the same bits
    o2 = make_oracle(g)
    o2.solve(oracle_params(), 77)
    with GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, one_xcd=0) as r:
        r.step(p, 77)
        assert r.info("persist_used") == 1 and r.info("one_xcd_used") == 0
        assert_bit_equal(r.download()[0], o2.x, "x, one_xcd = 0")


def test_one_xcd_frame_stream(gpu):
    """TUM-sized frames through the graph sync on one handle (the facade's options): every frame solved once, on 32 tiles or
    fewer -- one XCD --, the oracle's bits."""
    from oracle import COracle
    from oracle.cbind import SyncParams as OSync, graph_sync as oracle_sync
    p, sp = default_params(), default_sync_params()
    used = 0
    with GraphRegularizer.empty(device=0, tile_single_max=640, stream_depth=5) as r:
        for k in range(12):
            g = graphgen.dataset_shaped(640, 480, 16, seed=40 + k) if k % 3 else graphgen.synthetic(900 + 30 * k, seed=k)
            var = np.full(g.V, 1e-4, np.float32)
            r.sync_features(g.pos, g.z, var, g.tris, sp)
            r.step(p, 60)
            used += r.info("one_xcd_used")
            s = oracle_sync(OSync(0, 0, 1, 0.01), g.pos, g.z, var, g.tris, None)
            o = COracle(g.pos, s["edges"], s["alpha"], s["beta"], s["z"], s["wgt"], x0=s["x0"])
            o.solve(oracle_params(), 60)
            assert_bit_equal(r.download()[0], o.x, "frame %d" % k)
        assert r.info("persist_recovered") == 0
    assert used >= 10, used


def test_a_one_xcd_launch_that_gives_up_drops_the_mode(gpu):
    """Hooks library: every resident launch counts as failed -- the solve is repeated by launches (the oracle's bits), and no
    later handle of the process tries one XCD again."""
    code = r'''
import numpy as np, sys
sys.path.insert(0, %r)
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params
from oracle import COracle
from oracle.cbind import default_params as oparams
g = graphgen.dataset_shaped(640, 480, 16)
p = default_params()
o = COracle(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt); o.solve(oparams(), 80)
r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0)
assert r.info("num_tiles") == 32
r.step(p, 80, sync=False)
assert r.info("one_xcd_used") == 1
x = r.download()[0]
assert r.info("persist_recovered") == 1
assert np.array_equal(x.view(np.uint32), o.x.view(np.uint32))
r.close()
assert _hl.load().flame_hip_test_hook(b"persist_fail", 0) == 0   # (the hook off again: what follows are honest launches)
r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0)
for _ in range(40):   # (sit out the back-off of the give-up: launches meanwhile; more iterations than the halo depth, or the
    r.step(p, 12, sync=False)  # solve never asks for the lease)
r.sync()
assert r.info("one_xcd_used") == 0
r.close()
r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0)
assert r.info("num_tiles") == 50, r.info("num_tiles")   # (resident again, but sized for all XCDs)
r.step(p, 80, sync=False)
assert r.info("persist_used") == 1 and r.info("one_xcd_used") == 0
assert np.array_equal(r.download()[0].view(np.uint32), o.x.view(np.uint32))
r.close()
print("dropped ok")
''' % ROOT
    out = subprocess.run([sys.executable, "-c", with_hooks(code, persist_fail=1)], env=hooks_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "dropped ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
