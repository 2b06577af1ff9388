"""Writes the plain-text scene tools/pin_upstream/dump_upstream.cc reads, from this repository's
synthetic graphs (flame_ros_amd/graphgen.py): `python tools/pin_upstream/make_scene.py tum scene.txt`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flame_ros_amd import graphgen  # noqa: E402


def write_scene(path, g, x0=None):
    x0 = g.z if x0 is None else x0
    with open(path, "w") as f:
        f.write("%d %d\n" % (g.V, g.E))
        for v in range(g.V):
            f.write("%.9g %.9g %.9g %.9g %.9g\n" % (g.pos[v, 0], g.pos[v, 1], g.z[v], g.wgt[v], x0[v]))
        for e in range(g.E):
            f.write("%d %d %.9g %.9g\n" % (g.edges[e, 0], g.edges[e, 1], g.alpha[e], g.beta[e]))


if __name__ == "__main__":
    g, _ = graphgen.named(sys.argv[1] if len(sys.argv) > 1 else "tum")
    write_scene(sys.argv[2] if len(sys.argv) > 2 else "scene.txt", g)
