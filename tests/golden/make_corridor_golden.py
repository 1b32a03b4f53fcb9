#!/usr/bin/env python
"""Golden bytes for the corridor wire format: one synthetic recorded corridor
serialised by an independent packer that follows the ROS 1 serialisation rules for msgs/corridor
(msgs/msg/corridor.msg, polyhedron.msg, facet3.msg: little endian, uint32 array lengths,
geometry_msgs/Vector3 = 3 x float64).  No ROS installation or recorded bag exists here: PARITY UNPINNED.
Writes tests/golden/corridor_msg.bin and corridor_msg.npz (the arrays it encodes)."""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def ros1_corridor(path_id, n_planes, planes, seeds, centers):
    out = struct.pack("<iI", path_id, len(n_planes))
    for k, m in enumerate(n_planes):
        out += struct.pack("<3d", *centers[k]) + struct.pack("<3d", *seeds[k]) + struct.pack("<I", int(m))
        for j in range(int(m)):
            out += struct.pack("<4d", *planes[k, j])
    return out


def main():
    from direct_amd import problems
    b = problems.make_batch("corridor", 1, 64, seed=4242)      # one 64-polytope corridor of the config-3 generator
    n_planes, planes, seeds = b.n_planes[0], b.planes[0], b.seeds[0]
    rng = np.random.default_rng(7)
    centers = seeds + rng.uniform(-0.3, 0.3, seeds.shape)       # polytope centers near their seeds, inside the boxes
    msg = ros1_corridor(17, n_planes, planes, seeds, centers)
    open(os.path.join(HERE, "corridor_msg.bin"), "wb").write(msg)
    np.savez_compressed(os.path.join(HERE, "corridor_msg.npz"), path_id=17, n_planes=n_planes, planes=planes, seeds=seeds,
                        centers=centers)
    print(len(msg), "bytes", int(n_planes.sum()), "facets")


if __name__ == "__main__":
    main()
