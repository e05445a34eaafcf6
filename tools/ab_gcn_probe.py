#!/usr/bin/env python3
"""Interleaved in-process A/B of the GCN stage (clip shapes, batch 256): default vs a diagnostic switch (e.g. CSK_GCN16=2: the
16-wide tile family forced)."""
import os, sys, statistics
os.environ["CSK_DIAG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _bootstrap
pkg = _bootstrap.load()
var, _, val = sys.argv[1].partition("=")
val = val or "1"
dev = "cuda:0"; A = pkg.ntu_graph().A
for (ci, co, t) in [(3, 64, 300), (64, 64, 300), (64, 128, 300), (128, 128, 150), (128, 256, 150), (256, 256, 75)]:
    g = pkg.GraphConvolution(ci, co, A).eval().to(dev)
    x = torch.rand(512, ci, t, 25, device=dev)
    res = {0: [], 1: []}; outs = {}
    for rnd in range(12):
        for flag in (0, 1):
            if flag: os.environ[var] = val
            else: os.environ.pop(var, None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); y = g(x); e1.record(); torch.cuda.synchronize()
            outs[flag] = y
            if rnd >= 2: res[flag].append(e0.elapsed_time(e1))
    os.environ.pop(var, None)
    m0, m1 = statistics.median(res[0]), statistics.median(res[1])
    same = torch.equal(outs[0], outs[1])
    print(f"{ci:3d}->{co:<3d}: default {m0:.3f} ms | {var}={val} {m1:.3f} ms | default/alt {m0/m1:.3f}  bitwise-equal {same}")
