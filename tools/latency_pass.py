#!/usr/bin/env python3
"""Workload of the few-stream latency trace: CoST-GCN, `--streams` streams (default 1), one frame per forward_step call,
latency mode, steady state (rocprofv3 --kernel-trace --stats target; bench.py's `costgcn_online.latency` leg times the same
calls with a host synchronisation per frame).
usage: python tools/latency_pass.py [--streams 1] [--frames 400] [--no-latency-mode]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import _bootstrap  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=1)
ap.add_argument("--frames", type=int, default=400)
ap.add_argument("--no-latency-mode", action="store_true")
args = ap.parse_args()
pkg = _bootstrap.load()
dev = torch.device("cuda:0")
net = pkg.CoStGcn(pkg.ntu_graph().A).eval()
bench.randomise_(net, seed=0)
net = net.to(dev)
if not args.no_latency_mode:
    net.set_latency_mode(8)
x = torch.rand((8, args.streams, 3, 25, 2), device=dev, generator=torch.Generator(device=dev).manual_seed(5))
for t in range(76 + 4 * 56):
    net.forward_step(x[t % 8])
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(args.frames):
    net.forward_step(x[t % 8])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.frames
print(f"{args.streams} stream(s): {dt * 1e3:.4f} ms per frame (pipelined), split_k {[net.layers[f'layer{i + 1}']._state.ksplit for i in range(10)]}")
