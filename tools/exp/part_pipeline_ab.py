"""Dev aid: the library's partition mode on ONE GPU (world 1, every record a send / receive of the rank with itself) with the
exchange pipelined behind the parts (option "pipeline") and without: iterations/s, exchange share.  python tools/exp/part_pipeline_ab.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flame_ros_amd import graphgen, partition  # noqa: E402
from flame_ros_amd.regularizer import default_params  # noqa: E402

p = default_params()
with partition.Communicator(0, 0, 1, partition.unique_id()) as comm:
    cfgs = [("50k", 2, 16), ("50k", 4, 16), ("200k", 8, 16)] if len(sys.argv) < 2 else [("50k", 2, 32), ("50k", 2, 48), ("200k", 8, 24), ("200k", 8, 32), ("200k", 8, 48), ("200k", 16, 32)]
    for name, k, depth in cfgs:
        g, it = graphgen.named(name)
        ref = None
        for pipe in (0, 1):
            with partition.Partition(comm, g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, parts_per_rank=k, halo_depth=depth) as ps:
                ps.set_option("pipeline", pipe)
                ps.step(p, it); ps.step(p, it); ps.sync()
                t0 = time.perf_counter()
                for _ in range(5):
                    ps.step(p, it)
                ps.sync()
                dt = (time.perf_counter() - t0) / 5
                x = ps.gather_solution()[0]
                if ref is None:
                    ref = x
                same = np.array_equal(x.view(np.uint32), ref.view(np.uint32))
                print("%-5s %d parts on one rank, halo depth %d, pipeline %d: %.3f ms per %d iterations = %.3f us/it, exchanges %d (pipelined %d), resident launches %d, same bits %s" % (
                    name, k, depth, pipe, dt * 1e3, it, dt * 1e6 / it, ps.info("exchanges"), ps.info("exchanges_pipelined"), ps.info("persist_launches", 0), same), flush=True)
