"""Row a7 (graph sync): the library's flame_hip_graph_sync against the oracle's statement of it
(oracle/nltgv2_oracle.c nltgv2_graph_sync), on plan-only handles (no GPU needed): derived edge
list, alpha, data terms, weights, initial x and scale, bit for bit; the variance gate; the error
conventions.  The GPU leg (solve on the synced graph, un-scaling) is in tests/test_gpu_sync.py.
Parameters: reference src/flame_offline_tum.cc:234-249, cfg/flame_offline_tum.yaml:87-92."""
import numpy as np
import pytest

from flame_ros_amd import lib
from flame_ros_amd.regularizer import GraphRegularizer, default_sync_params, feature_gate, FlameHipError
from oracle.cbind import SyncParams as OSync, feature_gate as oracle_gate, graph_sync as oracle_sync
from tests.util import assert_bit_equal, graphgen


def features(V, seed):
    g = graphgen.synthetic(V, seed=seed)
    rng = np.random.default_rng(seed)
    var = rng.uniform(1e-5, 9e-3, g.V).astype(np.float32)
    pred = (g.z + rng.normal(0, 0.01, g.V)).astype(np.float32)
    pred[rng.random(g.V) < 0.2] = np.nan  # new features have no prediction
    return g, var, pred


@pytest.mark.parametrize("adaptive,rescale,init_pred", [(0, 0, 1), (1, 0, 0), (0, 1, 1), (1, 1, 1)])
def test_sync_matches_oracle(adaptive, rescale, init_pred):
    g, var, pred = features(4000, 11)
    sp = default_sync_params(adaptive, rescale, init_pred, 0.01)
    osp = OSync(adaptive, rescale, init_pred, 0.01)
    want = oracle_sync(osp, g.pos, g.z, var, g.tris, pred)
    r = GraphRegularizer.empty(device=-1)
    scale = r.sync_features(g.pos, g.z, var, g.tris, sp, prediction=pred)
    assert r.E == len(want["edges"]) == g.E and r.info("V") == g.V and r.info("T") == g.T
    assert np.array_equal(r.edges(), want["edges"])
    assert np.array_equal(r.edges(), g.edges)  # = graphgen's i<j lexicographic edge list
    assert_bit_equal(r.plan_array("sync_alpha", np.float32), want["alpha"], "alpha")
    assert_bit_equal(want["alpha"], g.alpha, "alpha vs graphgen")
    assert_bit_equal(r.plan_array("sync_z", np.float32), want["z"], "z")
    assert_bit_equal(r.plan_array("sync_wgt", np.float32), want["wgt"], "wgt")
    assert_bit_equal(r.plan_array("sync_x0", np.float32), want["x0"], "x0")
    assert np.float32(scale) == np.float32(want["scale"])
    if rescale:
        assert abs(float(np.mean(want["z"], dtype=np.float64)) - 1.0) < 1e-5 and scale != 1.0
    else:
        assert scale == 1.0
    if not init_pred:
        assert_bit_equal(want["x0"], want["z"], "x0 = z without prediction")
    r.close()


@pytest.mark.parametrize("rule,ag,bg", [(1, 0.0, 0.0), (2, 0.0, 0.5), (3, 2.0, 0.0), (0, 1.0, 3.0)])
def test_sync_upstream_recall_switches(rule, ag, bg):
    """The [UPSTREAM-RECALL] edge-weight switches (include/flame_hip.h flame_hip_sync_params tail):
    alternatives to alpha = beta = 1/len, identical in the library and in the oracle."""
    g, var, pred = features(1500, 5)
    want = oracle_sync(OSync(0, 0, 1, 0.01, rule, ag, bg), g.pos, g.z, var, g.tris, pred)
    r = GraphRegularizer.empty(device=-1)
    r.sync_features(g.pos, g.z, var, g.tris, default_sync_params(0, 0, 1, 0.01, rule, ag, bg), prediction=pred)
    assert_bit_equal(r.plan_array("sync_alpha", np.float32), want["alpha"], "alpha")
    assert_bit_equal(r.plan_array("sync_beta", np.float32), want["beta"], "beta")
    inv = g.alpha
    a = (np.ones_like(inv) if rule in (1, 3) else inv) * np.float32(ag if ag else 1.0)
    b = (np.ones_like(inv) if rule in (1, 2) else inv) * np.float32(bg if bg else 1.0)
    assert_bit_equal(want["alpha"], a, "alpha rule")
    assert_bit_equal(want["beta"], b, "beta rule")
    # the solver sees them: t_ew / ew of the plan carry (alpha, beta)
    ew = r.plan_array("ew", np.float32).reshape(-1, 4)
    e_i2o = r.plan_array("e_i2o", np.int32)
    assert_bit_equal(ew[:, 0], want["alpha"][e_i2o], "plan alpha")
    assert_bit_equal(ew[:, 1], want["beta"][e_i2o], "plan beta")
    r.close()
    with pytest.raises(FlameHipError):
        GraphRegularizer.empty(device=-1).sync_features(g.pos, g.z, var, g.tris, default_sync_params(edge_weight_rule=7))


def test_d_sign_option_flips_the_edge_vectors():
    g = graphgen.synthetic(800, seed=2)
    ews = []
    for ds in (1, -1):
        r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, device=-1, d_sign=ds)
        ews.append(r.plan_array("ew", np.float32).reshape(-1, 4))
        r.close()
    assert_bit_equal(ews[0][:, :2], ews[1][:, :2], "weights")
    assert_bit_equal(ews[0][:, 2:], -ews[1][:, 2:], "d")


def test_feature_gate():
    var = np.float32([0.0, 0.00999, 0.01, 0.02, np.inf, np.nan, 1e-9])
    keep = feature_gate(var, 0.01)
    assert keep.tolist() == [True, True, False, False, False, False, True]
    assert np.array_equal(keep, oracle_gate(var, 0.01))


def test_sync_error_conventions():
    g, var, pred = features(500, 3)
    r = GraphRegularizer.empty(device=-1)
    bad = var.copy(); bad[7] = 0.5                      # fails the gate -> ERR_ARG
    with pytest.raises(FlameHipError) as e:
        r.sync_features(g.pos, g.z, bad, g.tris, default_sync_params())
    assert e.value.code == lib.ERR_ARG
    bad = var.copy(); bad[7] = 0.0                      # adaptive weight 1/0 -> non-finite input
    with pytest.raises(FlameHipError) as e:
        r.sync_features(g.pos, g.z, bad, g.tris, default_sync_params(adaptive_data_weights=True))
    assert e.value.code == lib.ERR_NAN
    mu = g.z.copy(); mu[3] = np.nan
    with pytest.raises(FlameHipError) as e:
        r.sync_features(g.pos, mu, var, g.tris, default_sync_params())
    assert e.value.code == lib.ERR_NAN
    t = g.tris.copy(); t[0, 1] = g.V                    # triangle index out of range
    with pytest.raises(FlameHipError) as e:
        r.sync_features(g.pos, g.z, var, t, default_sync_params())
    assert e.value.code == lib.ERR_ARG
    # the handle survives the failures: a good frame afterwards works, sizes follow the frame
    r.sync_features(g.pos, g.z, var, g.tris, default_sync_params())
    assert (r.info("V"), r.info("E"), r.info("T")) == (g.V, g.E, g.T)
    g2, var2, _ = features(900, 4)
    r.sync_features(g2.pos, g2.z, var2, g2.tris, default_sync_params())
    assert (r.info("V"), r.info("E"), r.info("T")) == (g2.V, g2.E, g2.T)
    # an edge list is only available for graphs the library derived itself
    r.reupload(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris)
    with pytest.raises(FlameHipError):
        r.edges()
    r.close()


def test_sync_without_triangles_and_empty():
    r = GraphRegularizer.empty(device=-1)
    pos = np.float32([[1, 2], [30, 40], [50, 5]])
    s = r.sync_features(pos, np.float32([0.5, 0.6, 0.7]), np.float32([1e-4] * 3), np.zeros((0, 3), np.int32),
                        default_sync_params())
    assert s == 1.0 and (r.info("V"), r.info("E"), r.info("T")) == (3, 0, 0)
    s = r.sync_features(np.zeros((0, 2), np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32),
                        np.zeros((0, 3), np.int32), default_sync_params(rescale_data=True))
    assert s == 1.0 and r.info("V") == 0
    r.close()
