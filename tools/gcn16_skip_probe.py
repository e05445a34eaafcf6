#!/usr/bin/env python3
"""Diagnostic: time the step-shape graph-conv launches of CoST-GCN (1024 streams) with the aggregation phase or the MFMA phase
of gcn16_kernel switched off (CSK_GCN16_SKIP = 1 / 2 under CSK_DIAG=1; results are wrong in those runs)."""
import os, sys, statistics
os.environ["CSK_DIAG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _bootstrap, bench
pkg = _bootstrap.load()
dev = torch.device("cuda:0")
A = pkg.ntu_graph().A
P = 1024 * 2 * 25
for (ci, co, frames) in [(64, 64, 4), (128, 128, 2), (256, 256, 1), (128, 256, 2)]:
    g = pkg.GraphConvolution(ci, co, A).eval().to(dev)
    x = torch.rand((frames, ci, P), device=dev)
    y = torch.empty((frames, co, P), device=dev)
    res = {}
    for skip in (0, 1, 2, 0):
        os.environ["CSK_GCN16_SKIP"] = str(skip)
        ts = []
        for it in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.stage(x, y, n_seg=frames, frames=2048, x_strides=(ci * P, P), y_strides=(co * P, P))
            e1.record()
            torch.cuda.synchronize()
            if it >= 5:
                ts.append(e0.elapsed_time(e1))
        res[skip] = statistics.median(ts)
    print(f"GCN16_SKIP {ci}->{co} x{frames} frames: full {res[0] * 1e3:.1f} us | no aggregation {res[1] * 1e3:.1f} us | no MFMA {res[2] * 1e3:.1f} us")
