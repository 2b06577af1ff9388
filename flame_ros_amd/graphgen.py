"""Synthetic Delaunay vertex graphs shaped like FLaME's feature graphs.

Recipe: SURVEY.md section 8(d) / BASELINE.md section 2 (configs 2, 4, 5 of BASELINE.json):
V i.i.d. uniform pixel positions, Delaunay triangulation, unique undirected edges oriented i<j,
alpha = beta = 1/|pos_i - pos_j|, piecewise-planar idepth data with noise and outliers,
data_weight = 1.  In FLaME proper the graph comes from Flame::update (feature detection +
triangulation, reference src/flame_offline_tum.cc:578); the detection grid win_size
(cfg/flame_offline_tum.yaml:78) sets V ~ (W/win)*(H/win), which `dataset_shaped` mimics.
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class SyntheticGraph:
    width: int
    height: int
    pos: np.ndarray      # [V,2] float32 pixel coords
    edges: np.ndarray    # [E,2] int32, oriented i -> j with i < j
    tris: np.ndarray     # [T,3] int32
    alpha: np.ndarray    # [E] float32
    beta: np.ndarray     # [E] float32
    z: np.ndarray        # [V] float32 data term (noisy idepth)
    wgt: np.ndarray      # [V] float32 data weight

    @property
    def V(self):
        return int(self.pos.shape[0])

    @property
    def E(self):
        return int(self.edges.shape[0])

    @property
    def T(self):
        return int(self.tris.shape[0])


def edges_from_triangles(tris):
    """Unique undirected edges of a triangle list, oriented i<j, sorted lexicographically."""
    t = np.asarray(tris, dtype=np.int64)
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]], axis=0)
    e.sort(axis=1)
    e = np.unique(e, axis=0)
    return e.astype(np.int32)


def idepth_field(pos, width, rng, noise=0.02, outlier_frac=0.05):
    u, v = pos[:, 0].astype(np.float64), pos[:, 1].astype(np.float64)
    z = 0.5 + 0.001 * u - 0.0005 * v + 0.3 * (u > width / 2)
    z = z + rng.normal(0.0, noise, size=len(u))
    out = rng.random(len(u)) < outlier_frac
    z = z + out * rng.uniform(-0.5, 0.5, size=len(u))
    return np.maximum(z, 0.01).astype(np.float32)


def from_points(pos, width, height, rng):
    from scipy.spatial import Delaunay  # same image here and on the GPU box

    pos = np.ascontiguousarray(pos, dtype=np.float32)
    tris = Delaunay(pos.astype(np.float64)).simplices.astype(np.int32)
    edges = edges_from_triangles(tris)
    d = pos[edges[:, 0]] - pos[edges[:, 1]]
    length = np.sqrt((d.astype(np.float32) ** 2).sum(1, dtype=np.float32))
    alpha = (np.float32(1.0) / length).astype(np.float32)
    z = idepth_field(pos, width, rng)
    return SyntheticGraph(width, height, pos, edges, tris, alpha, alpha.copy(), z,
                          np.ones(len(z), np.float32))


def synthetic(num_vertices, width=640, height=480, seed=0):
    """BASELINE configs 2/4 (640x480) and 5 (1280x1024): uniform random feature positions."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pos = rng.random((num_vertices, 2)) * np.array([width, height])
    return from_points(pos.astype(np.float32), width, height, rng)


def dataset_shaped(width, height, win_size, seed=0):
    """One jittered feature per win_size x win_size detection cell (configs 1 and 3 stand-ins:
    TUM 640x480 @16 -> V=1200; EuRoC 752x480 @8 -> V=5640)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    gx, gy = np.meshgrid(np.arange(width // win_size), np.arange(height // win_size))
    cell = np.column_stack([gx.ravel(), gy.ravel()]).astype(np.float64)
    pos = (cell + 0.1 + 0.8 * rng.random(cell.shape)) * win_size
    return from_points(pos.astype(np.float32), width, height, rng)


NAMED = {
    "5k": dict(num_vertices=5000, width=640, height=480, iters=200),
    "50k": dict(num_vertices=50000, width=640, height=480, iters=500),
    "200k": dict(num_vertices=200000, width=1280, height=1024, iters=500),
}


# dataset-shaped stand-ins for BASELINE configs 1 and 3 (the datasets themselves are not available:
# no network, no image decoder): one jittered feature per detection cell
GRID = {
    "tum": dict(width=640, height=480, win=16, iters=200),     # fr3/structure_texture_far, V = 1200
    "euroc": dict(width=752, height=480, win=6, iters=200),    # V1_01, "~10k vertices": V = 10000
}


def named(name, seed=0):
    if name in GRID:
        c = GRID[name]
        return dataset_shaped(c["width"], c["height"], c["win"], seed), c["iters"]
    if name[0] == "v" and name[1:].isdigit():  # "v2000": that many uniform vertices on 640 x 480, 200 iterations
        return synthetic(int(name[1:]), 640, 480, seed), 200
    if name[0] == "g" and name[1:].isdigit():  # "g24": one jittered feature per 24-pixel cell of 640 x 480
        return dataset_shaped(640, 480, int(name[1:]), seed), 200
    c = NAMED[name]
    return synthetic(c["num_vertices"], c["width"], c["height"], seed), c["iters"]
