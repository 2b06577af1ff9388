"""-m gpu: the stat key `coverage` (reference src/utils.cc:122) and the four debug images of
flame::Flame (reference src/flame_offline_tum.cc:731-766) rendered on the device, byte for byte
against the oracle's statement (oracle/nltgv2_oracle.c nltgv2_coverage / nltgv2_debug_image)."""
import numpy as np
import pytest

from flame_ros_amd import lib
from flame_ros_amd.regularizer import GraphRegularizer, default_params, default_tri_params
from oracle.cbind import (TriParams as OTri, coverage as oracle_coverage, debug_image as oracle_image,
                          depthmaps as oracle_depthmaps, triangles as oracle_triangles)
from tests.util import graphgen, make_oracle, oracle_params

pytestmark = pytest.mark.gpu

K = np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]])
KINV = np.linalg.inv(K).astype(np.float32)


def setup(g, iters, opts, poke=True):
    o = make_oracle(g)
    r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, **opts)
    o.solve(oracle_params(), iters)
    r.step(default_params(), iters)
    if poke:  # invalid idepths: NaN / negative vertices must behave the same on both sides
        x = o.x.copy()
        x[::97] = np.nan
        x[11::89] = -0.5
        o.set_state(x=x)
        r.set_state(x=x)
    return o, r


@pytest.mark.parametrize("shape", ["tum", "random3k", "wide"])
def test_coverage_and_debug_images(gpu, shape):
    if shape == "tum":
        g, W, H, opts, scale = graphgen.dataset_shaped(640, 480, 16, seed=3), 640, 480, dict(tile_own=4096), 1.0
    elif shape == "random3k":
        g, W, H, opts, scale = graphgen.synthetic(3000, seed=5), 640, 480, {}, 0.7
    else:  # image smaller than the feature extent: lines and squares are clipped per pixel
        g, W, H, opts, scale = graphgen.synthetic(1500, seed=6), 500, 300, {}, 1.9
    o, r = setup(g, 30, opts)
    tp = default_tri_params(W, H)
    otp = OTri(*[getattr(tp, f[0]) for f in tp._fields_])
    _, tv_o, vn_o = oracle_triangles(otp, KINV, g.pos, o.x, g.tris)
    idm_o, _, _ = oracle_depthmaps(W, H, g.pos, o.x, g.tris, tv_o, True, KINV, 0.1, 100.0)
    # coverage rides on frame_results' single synchronisation
    out = r.frame_results(default_params(), KINV, tp, with_coverage=True)
    cov = out[-1]
    assert np.float32(cov) == np.float32(oracle_coverage(idm_o)), (cov, oracle_coverage(idm_o))
    assert 0.3 < cov < 1.0
    # raw features: a superset of the vertices, some of them outside the image
    rng = np.random.default_rng(1)
    fpos = np.concatenate([g.pos, rng.uniform(-5, [W + 5, H + 5], (200, 2)).astype(np.float32)])
    fmu = np.concatenate([g.z, rng.uniform(0.0, 2.5, 200).astype(np.float32)])
    for kind, name in ((lib.IMG_WIREFRAME, "wireframe"), (lib.IMG_FEATURES, "features"),
                       (lib.IMG_NORMALS, "normals"), (lib.IMG_IDEPTHMAP, "idepthmap")):
        want = oracle_image(kind, W, H, scale, g.pos, o.x, g.tris, tv_o, vn_o, idm_o, fpos, fmu)
        got = r.debug_image(kind, KINV, tp, scale, fpos, fmu)
        bad = (got != want).any(axis=2)
        assert not bad.any(), "%s: %d pixels differ, first at %s" % (name, int(bad.sum()), np.argwhere(bad)[:3].tolist())
        assert (want.sum(axis=2) > 0).mean() > 0.01, name  # something was drawn
    # the dense map getter after the images reuses the same raster
    idm, _, _ = r.depthmaps(KINV, tp, filtered=True, cloud=False)
    assert np.array_equal(np.isnan(idm), np.isnan(idm_o))
    r.close()


def test_images_follow_the_state(gpu):
    """The raster cache is keyed on the solver state: more iterations => a new image."""
    g = graphgen.dataset_shaped(640, 480, 16, seed=9)
    o, r = setup(g, 5, {}, poke=False)
    tp = default_tri_params(640, 480)
    otp = OTri(*[getattr(tp, f[0]) for f in tp._fields_])
    a = r.debug_image(lib.IMG_IDEPTHMAP, KINV, tp)
    assert np.array_equal(a, r.debug_image(lib.IMG_IDEPTHMAP, KINV, tp))
    o.solve(oracle_params(), 60)
    r.step(default_params(), 60)
    _, tv_o, vn_o = oracle_triangles(otp, KINV, g.pos, o.x, g.tris)
    idm_o, _, _ = oracle_depthmaps(640, 480, g.pos, o.x, g.tris, tv_o, True, KINV, 0.1, 100.0)
    b = r.debug_image(lib.IMG_IDEPTHMAP, KINV, tp)
    assert np.array_equal(b, oracle_image(lib.IMG_IDEPTHMAP, 640, 480, 1.0, g.pos, o.x, g.tris, tv_o, vn_o, idm_o))
    assert not np.array_equal(a, b)
    # no triangles: nothing covered, black images, coverage 0
    r2 = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=None)
    r2.step(default_params(), 3)
    assert r2.frame_results(default_params(), KINV, tp, with_coverage=True)[-1] == 0.0
    assert not r2.debug_image(lib.IMG_WIREFRAME, KINV, tp).any()
    r2.close()
    r.close()
