"""-m gpu: the halo pack/unpack kernels and HipSubdomainSolver on ONE GPU: a graph is cut into
`world` subdomains that all live on cuda:0 and exchange their halos through device buffers inside
this process (what RCCL moves between GPUs at N>1).  Result: bit-identical to the oracle."""
import numpy as np
import pytest

from flame_ros_amd import dist as fdist
from flame_ros_amd.regularizer import default_params
from oracle import COracle
from oracle.cbind import default_params as oracle_params
from tests.halo_driver import run_subdomains_one_gpu
from tests.util import graphgen

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,depth,iters", [(2, 4, 22), (4, 3, 10)])
def test_subdomains_on_one_gpu(gpu, world, depth, iters):
    g = graphgen.synthetic(6000, seed=21)
    subs, solvers = run_subdomains_one_gpu(g, world, depth, iters)
    o = COracle(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt)
    o.solve(oracle_params(), iters)
    for r, s in enumerate(subs):
        x, w1, w2, q = solvers[r].download()
        own = slice(0, s.n_own)
        assert np.array_equal(x[own].view(np.uint32), o.x[s.vid[own]].view(np.uint32)), r
        assert np.array_equal(w1[own].view(np.uint32), o.w1[s.vid[own]].view(np.uint32)), r
        oe = np.flatnonzero(s.e_owned)
        assert np.array_equal(q[oe].view(np.uint32), o.q[s.eid[oe]].view(np.uint32)), r
