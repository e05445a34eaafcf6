#!/usr/bin/env python3
"""Interleaved in-process A/B of the fused first block (csk_block_few_channels_f32) against its two launches: ST-GCN clip forward
at batch 256 (and the A-GCN-free Kinetics-shape ST-GCN), plus the block alone under HIP events."""
import os, sys, statistics, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, _bootstrap, bench
pkg = _bootstrap.load()
dev = "cuda:0"
net = pkg.StGcn(pkg.ntu_graph().A).eval(); bench.randomise_(net, 0); net = net.to(dev)
x = torch.rand((256, 3, 300, 25, 2), device=dev)
blk = net.layers.layer1
res = {True: [], False: []}
for rnd in range(8):
    for fuse in (True, False):
        pkg.SpatioTemporalBlock.fuse_few_channels = fuse
        net(x); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): net(x)
        torch.cuda.synchronize()
        if rnd >= 2: res[fuse].append((time.perf_counter() - t0) / 3 * 1e3)
print(f"FIRST_BLOCK clip forward b256: fused {statistics.median(res[True]):.3f} ms | two launches {statistics.median(res[False]):.3f} ms")
xb = torch.rand((512, 3, 300, 25), device=dev)
rb = {True: [], False: []}
for rnd in range(10):
    for fuse in (True, False):
        pkg.SpatioTemporalBlock.fuse_few_channels = fuse
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); blk(xb); e1.record(); torch.cuda.synchronize()
        if rnd >= 2: rb[fuse].append(e0.elapsed_time(e1))
print(f"FIRST_BLOCK block 1 alone (512 sequences x 300 frames): fused {statistics.median(rb[True]):.3f} ms | two launches {statistics.median(rb[False]):.3f} ms")
