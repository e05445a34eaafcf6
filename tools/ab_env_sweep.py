#!/usr/bin/env python3
"""Interleaved in-process sweep of diagnostic env settings on the online path (CoST-GCN, 1024 streams, one stream shard by
default): every configuration is timed in turn, several rounds, medians reported.  The library reads the switches per launch.
usage: python tools/ab_env_sweep.py [--shards 1] [--model costgcn|coagcn] "CSK_GCN16_STAGGER=1" "CSK_GCN16_STAGGER=33,CSK_TCN16_STAGGER=33" ...
(the empty string "" is the default configuration)"""
import argparse, os, statistics, sys, time
os.environ["CSK_DIAG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _bootstrap, bench
ap = argparse.ArgumentParser()
ap.add_argument("--shards", type=int, default=1)
ap.add_argument("--streams", type=int, default=1024)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--cycles", type=int, default=16)
ap.add_argument("--model", default="costgcn", choices=["costgcn", "coagcn"])
ap.add_argument("configs", nargs="*", default=[""])
a = ap.parse_args()
pkg = _bootstrap.load()
from continual_skeletons_amd import parallel
dev = torch.device("cuda:0")
V = 18 if a.model == "coagcn" else 25


def make():
    if a.model == "coagcn":
        net = pkg.CoAGcn(pkg.kinetics_graph().A, bench.KIN_SHAPE, 400).eval()
        bench.randomise_(net, seed=0, attn_scale=1 / 18)
    else:
        net = pkg.CoStGcn(pkg.ntu_graph().A).eval()
        bench.randomise_(net, seed=0)
    return net.to(dev)


eng = parallel.StreamShards(make, a.streams, a.shards, dev)
frames = torch.rand((8, a.streams, 3, V, 2), device=dev)
for t in range(80):
    eng.forward_cycle([frames[(4 * t + f) % 8] for f in range(4)])
configs = [dict(kv.split("=") for kv in c.split(",") if kv) for c in a.configs]
keys = sorted({k for c in configs for k in c})
res = [[] for _ in configs]
for rnd in range(a.rounds):
    for i, c in enumerate(configs):
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(c)
        for w in range(2):
            eng.forward_cycle([frames[f] for f in range(4)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for cyc in range(a.cycles):
            eng.forward_cycle([frames[(4 * cyc + f) % 8] for f in range(4)])
        torch.cuda.synchronize()
        if rnd >= 1:
            res[i].append((time.perf_counter() - t0) / a.cycles * 1e3)
base = statistics.median(res[0])
for c, r in zip(a.configs, res):
    m = statistics.median(r)
    print(f"AB_SWEEP {a.model} shards={a.shards} [{c or 'default'}]: {m:.4f} ms/cycle  {4 * a.streams / m:.0f} kframes/s  x{base / m:.4f} vs first  (min {min(r):.4f} max {max(r):.4f})")
