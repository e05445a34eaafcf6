#!/usr/bin/env python3
"""hipGraph capture of the ST-GCN clip forward (every launch goes through the C ABI on torch's current stream;
no allocation / sync inside the entry points): eager vs graph replay latency at small batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _bootstrap
pkg = _bootstrap.load()
import bench
dev = "cuda:0"
net = pkg.StGcn(pkg.ntu_graph().A).eval(); bench.randomise_(net, 0); net = net.to(dev)
for n in (1, 8, 64):
    x = torch.rand((n, 3, 300, 25, 2), device=dev)
    for _ in range(3): ref = net(x)                      # warm-up: folds weights, raises LDS caps, fills the allocator
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = net(x)
    g.replay(); torch.cuda.synchronize()
    assert torch.equal(out, ref), "graph replay differs from eager"
    def t(fn, it=30):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(it): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
    print(f"batch {n:3d}: eager {t(lambda: net(x)):7.3f} ms   graph replay {t(g.replay):7.3f} ms   (bitwise equal)")
