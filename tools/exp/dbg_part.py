"""r06: the parts of the 200 k graph cut 8-way on one rank -- local size, tile plan, resident launches (why half of them were
not resident before regular_next_attempt(): 511-515 tiles, four edges per thread, LDS + staging over the limit)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flame_ros_amd import graphgen, partition
from flame_ros_amd.regularizer import default_params
g, _ = graphgen.named("200k")
p = default_params()
with partition.Communicator(0, 0, 1, partition.unique_id()) as comm:
    with partition.Partition(comm, g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, parts_per_rank=8, halo_depth=16) as ps:
        ps.step(p, 100)
        ps.sync()
        for i in range(8):
            print(i, {k: ps.info(k, i) for k in ("n_own", "n_ext", "e_loc", "persist_launches", "num_tiles", "tile_depth", "tile_threads", "tile_ept",
                                                 "tile_lds_bytes", "tile_fat", "tile_imbalance_pct")})
