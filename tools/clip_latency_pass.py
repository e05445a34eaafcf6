#!/usr/bin/env python3
"""Small-batch clip latency: StGcn forward as a replayed hipGraph for a sweep of split_k, or (--trace) a few eager forwards
as a rocprofv3 target.  usage: python tools/clip_latency_pass.py [--batch 1] [--split-k 0,2,4,8] [--trace]"""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import _bootstrap  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--split-k", default="0,2,4,6,8,12,16")
ap.add_argument("--trace", action="store_true")
ap.add_argument("--gcn-split-k", type=int, default=-1)
ap.add_argument("--forwards", type=int, default=200)
ap.add_argument("--max-seq", type=int, default=None, help="max_sequences of set_latency_mode (default: the library's 6)")
a = ap.parse_args()
pkg = _bootstrap.load()
dev = "cuda:0"
net = pkg.StGcn(pkg.ntu_graph().A).eval()
bench.randomise_(net, 0)
net = net.to(dev)
x = torch.rand((a.batch, 3, 300, 25, 2), device=dev, generator=torch.Generator(device=dev).manual_seed(41))
for sk in [int(v) for v in a.split_k.split(",")]:
    net.set_latency_mode(sk, None if a.gcn_split_k < 0 else a.gcn_split_k, a.max_seq)
    for _ in range(3):
        out = net(x)
    torch.cuda.synchronize()
    if a.trace:
        for _ in range(20):
            out = net(x)
        torch.cuda.synchronize()
        print(f"CLIP_LATENCY_PASS trace batch={a.batch} split_k={sk}")
        continue
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        gout = net(x)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.forwards):
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"CLIP_LATENCY_PASS batch={a.batch} split_k={sk} gcn_split_k={a.gcn_split_k} max_seq={a.max_seq} graph_ms_p50={statistics.median(ts) * 1e3:.4f} "
          f"p99={sorted(ts)[int(0.99 * (len(ts) - 1))] * 1e3:.4f}")
