#!/usr/bin/env python
"""Golden vectors for the output-sampling step.

An independent NumPy restatement of the caller's sampling loop (teach_repeat_planner.cpp:1551-1566 over
utils/bezier_base.h:77-127), fed with the phase-1 outputs of the committed solver fixtures.  The original
cannot be run here (ROS / Eigen are absent), so like the solver fixtures these are restatement-generated:
PARITY UNPINNED.  Writes tests/golden/sample_<case>.npz."""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
C5 = np.array([1, 5, 10, 10, 5, 1.0])
C4 = np.array([1, 4, 6, 4, 1.0])
C3 = np.array([1, 3, 3, 1.0])


def sample_one(n_seg, bez, T, dt, capacity):
    if (T[:n_seg] < 0).any():
        return dict(count=-1)
    pos, vel, acc, seg_first = [], [], [], []
    length, pre = 0.0, None
    for i in range(n_seg):
        c = bez[i].reshape(3, 6)
        seg_first.append(len(pos))
        t, step = 0.0, dt / float(T[i])
        while t < 1.0:
            tj = np.array([math.pow(t, j) for j in range(6)])
            uj = np.array([math.pow(1.0 - t, j) for j in range(6)])
            p = np.array([sum(C5[j] * c[d, j] * tj[j] * uj[5 - j] for j in range(6)) for d in range(3)]) * float(T[i])
            v = np.array([sum(C4[j] * 5 * (c[d, j + 1] - c[d, j]) * tj[j] * uj[4 - j] for j in range(5)) for d in range(3)])
            a = np.array([sum(C3[j] * 5 * 4 * (c[d, j + 2] - 2 * c[d, j + 1] + c[d, j]) * tj[j] * uj[3 - j]
                              for j in range(4)) for d in range(3)]) / float(T[i])
            if pre is not None:
                length += float(np.sqrt(((pre - p) ** 2).sum()))
            pre = p
            pos.append(p); vel.append(v); acc.append(a)
            t += step
    n = len(pos)
    out = dict(count=n, seg_first=np.array(seg_first, np.int32), length=length,
               vmax=float(np.abs(np.array(vel)).max()), amax=float(np.abs(np.array(acc)).max()))
    for k, arr in (("pos", pos), ("vel", vel), ("acc", acc)):
        full = np.zeros((capacity, 3))
        full[:min(n, capacity)] = np.array(arr)[:capacity]
        out[k] = full
    return out


def main():
    for case, dt, cap in (("corridor_n8", 0.2, 256), ("free_n5", 0.1, 512), ("config1_n50", 0.2, 1024)):
        g = np.load(os.path.join(HERE, case + ".npz"))
        n_seg, bez, T = g["n_seg"], g["p1_bez"], g["p1_T"]
        B, nm = T.shape
        res = [sample_one(int(n_seg[b]), bez[b], T[b], dt, cap) for b in range(B)]
        seg_first = np.zeros((B, nm), np.int32)
        for b in range(B):
            seg_first[b, :n_seg[b]] = res[b]["seg_first"]
        np.savez_compressed(os.path.join(HERE, "sample_" + case + ".npz"), n_seg=n_seg, bez=bez, T=T, dt=dt, capacity=cap,
                            count=np.array([r["count"] for r in res], np.int32), seg_first=seg_first,
                            pos=np.stack([r["pos"] for r in res]), vel=np.stack([r["vel"] for r in res]),
                            acc=np.stack([r["acc"] for r in res]), length=np.array([r["length"] for r in res]),
                            vmax=np.array([r["vmax"] for r in res]), amax=np.array([r["amax"] for r in res]))
        print(case, "points", [r["count"] for r in res])


if __name__ == "__main__":
    sys.exit(main())
