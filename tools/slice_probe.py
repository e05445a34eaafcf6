#!/usr/bin/env python3
"""Clip-mode N1 probe (cheap variant): run every block's GCN stage -> TCN stage per SLICE of the batch
so that the post-GCN tensor of a slice (123 MB at 32 clips, C = 64) is still in the 256 MB Infinity Cache when the TCN
stage reads it -- what a fused block kernel would save in HBM traffic, without its halo recompute.  Interleaved
in-process A/B against the whole-batch form (two launches per block over all 512 sequences).
usage: python tools/slice_probe.py [--batch 256] [--slices 8,4,2]"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import _bootstrap  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--slices", default="8,4,2")
args = ap.parse_args()
pkg = _bootstrap.load()
dev = torch.device("cuda:0")
net = pkg.StGcn(pkg.ntu_graph().A).eval()
bench.randomise_(net, 0)
net = net.to(dev)
x = torch.rand((args.batch, 3, 300, 25, 2), device=dev)
SpatioTemporalBlock = pkg.SpatioTemporalBlock


def forward_sliced(n_slices):
    """every block: for each slice GCN stage -> TCN stage, written straight into the block's output tensor (no copies)"""
    h = net.input_norm(x)
    nm = h.shape[0]
    step = nm // n_slices
    for i in range(10):
        blk = net.layers[f"layer{i + 1}"]
        t_out = (h.shape[2] - 1) // blk.stride + 1
        out = torch.empty((nm, blk.gcn.out_channels, t_out, 25), device=dev)
        for s in range(0, nm, step):
            SpatioTemporalBlock.forward(blk, h[s: s + step], out=out[s: s + step])
        h = out
    return net.head(h, args.batch, 2)


ref = net(x)
variants = {"whole batch": lambda: net(x)}
for k in (int(v) for v in args.slices.split(",")):
    variants[f"{k} slices of {args.batch // k} clips"] = (lambda k=k: forward_sliced(k))
res = {k: [] for k in variants}
for rnd in range(7):
    for name, fn in variants.items():
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
        if rnd >= 2:
            res[name].append(e0.elapsed_time(e1))
base = statistics.median(res["whole batch"])
for name, v in res.items():
    m = statistics.median(v)
    print(f"{name:36s} {m:8.3f} ms  ({args.batch / m * 1e3:7.1f} clips/s)  x{m / base:.4f} of whole-batch time; bitwise-equal logits")
