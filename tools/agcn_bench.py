#!/usr/bin/env python3
"""BASELINE config 4: AGCN clip forward (Kinetics-400 shape, batch 64) and CoAGCN online step, vs ST-GCN / CoST-GCN
at the same shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _bootstrap
pkg = _bootstrap.load()
import bench
dev = "cuda:0"
A = pkg.kinetics_graph().A
shape = (3, 300, 18, 2)

def timeit(fn, iters):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters

x = torch.rand((64,) + shape, device=dev)
for name, cls in (("ST-GCN", pkg.StGcn), ("AGCN", pkg.AGcn)):
    net = cls(A, shape, 400).eval(); bench.randomise_(net, 0); net = net.to(dev)
    dt = timeit(lambda: net(x), 5)
    print(f"{name:8s} clip batch 64 Kinetics shape: {dt*1e3:8.2f} ms/step  {64/dt:8.1f} clips/s")
    del net
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
shards = int(sys.argv[2]) if len(sys.argv) > 2 else 2
from continual_skeletons_amd import parallel
frames = torch.rand((8, streams, 3, 18, 2), device=dev)
for name, cls in (("CoST-GCN", pkg.CoStGcn), ("CoAGCN", pkg.CoAGcn)):
    def make():
        net = cls(A, shape, 400).eval(); bench.randomise_(net, 0); return net.to(dev)
    eng = parallel.StreamShards(make, streams, shards, dev)
    for t in range(80): eng.forward_cycle([frames[t % 8]])
    i = [0]
    def cyc():
        eng.forward_cycle([frames[(i[0] + f) % 8] for f in range(4)]); i[0] += 4
    dt = timeit(cyc, 12)
    print(f"{name:8s} online {streams} streams ({shards} shards) Kinetics shape: {dt/4*1e3:8.3f} ms/frame-step  {4*streams/dt:10.0f} frames/s")
    del eng
