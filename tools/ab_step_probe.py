#!/usr/bin/env python3
"""Interleaved in-process A/B of a diagnostic env switch on the whole online path (CoST-GCN, 1024 streams, 2 shards).
usage: python tools/ab_step_probe.py CSK_SLOW_EPI"""
import os, sys, time, statistics
os.environ["CSK_DIAG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _bootstrap, bench
pkg = _bootstrap.load()
from continual_skeletons_amd import parallel
var, _, val = sys.argv[1].partition("=")
val = val or "1"
dev = torch.device("cuda:0")
streams = 1024


def make():
    net = pkg.CoStGcn(pkg.ntu_graph().A).eval()
    bench.randomise_(net, seed=0)
    return net.to(dev)


eng = parallel.StreamShards(make, streams, 2, dev)
frames = torch.rand((8, streams, 3, 25, 2), device=dev)
for t in range(76 + 4 * 55):
    eng.forward_cycle([frames[t % 8]])
res = {0: [], 1: []}
for rnd in range(10):
    for flag in (0, 1):
        if flag:
            os.environ[var] = val
        else:
            os.environ.pop(var, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c in range(12):
            eng.forward_cycle([frames[(4 * c + f) % 8] for f in range(4)])
        torch.cuda.synchronize()
        if rnd >= 2:
            res[flag].append(4 * streams * 12 / (time.perf_counter() - t0))
m0, m1 = statistics.median(res[0]), statistics.median(res[1])
print(f"online 1024 streams: default {m0:,.0f} frames/s | {var}={val} {m1:,.0f} frames/s | default/alt {m0 / m1:.4f}")
