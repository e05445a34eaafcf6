"""Workload of the PMC traffic passes for the online path (tools/profile.sh): ONLY 4-frame launches over all 1024
streams in one shard -- the launch shape bench.py's kernel-timing pass measures roofline.achieved on -- from a clean
state, so that the tcn_step_kernel launches of the run are the ten blocks in their steady 4:3:3 (64/128/256-channel)
proportion and no single-frame warm-up launches dilute the per-launch average."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import _bootstrap
import bench

pkg = _bootstrap.load()
dev = torch.device("cuda:0")
streams, cycles = 1024, 64
net = pkg.CoStGcn(pkg.ntu_graph().A).eval()
bench.randomise_(net, seed=0)
net = net.to(dev)
frames = torch.rand((8, streams, 3, 25, 2), device=dev)
for c in range(cycles):
    net.forward_cycle([frames[(4 * c + f) % 8] for f in range(4)])
torch.cuda.synchronize()
print(f"{cycles} cycles of 4 frames x {streams} streams")
