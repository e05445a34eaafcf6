#!/usr/bin/env python3
"""Workload of the online-path profile passes (tools/profile_r04.sh): EXACTLY bench.py's online shape --
CoST-GCN, 1024 streams, cycles of 4 frames, native plan -- from a clean state, warm-up cycles included (the trace
summariser tools/summarize_layers.py keeps the last --cycles cycles of every HIP stream, which are steady state:
all ten blocks emit, the temporal pool is full).
usage: python tools/online_pass.py [--shards 1|2] [--fpl 4] [--cycles 24] [--streams 1024]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import _bootstrap  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shards", type=int, default=1)
ap.add_argument("--fpl", type=int, default=4)
ap.add_argument("--cycles", type=int, default=24)
ap.add_argument("--streams", type=int, default=1024)
ap.add_argument("--warm-cycles", type=int, default=0, help="default: enough to fill the stack and the temporal pool")
ap.add_argument("--no-fuse", action="store_true", help="two launches per block everywhere (no csk_co_block_step_f32)")
ap.add_argument("--force-ksplit", type=int, default=0, help="split-K factor forced on blocks with C_out >= --ksplit-min-c (experiment)")
ap.add_argument("--ksplit-min-c", type=int, default=256)
ap.add_argument("--no-fuse-attention", action="store_true", help="A-GCN: embedding conv and attention as two launches (Python engine)")
ap.add_argument("--model", default="costgcn", choices=["costgcn", "coagcn", "coagcn_ntu"],
                help="coagcn: BASELINE configs[3], Kinetics-400 shape (V = 18); coagcn_ntu: the same model on the NTU-60 shape (V = 25)")
args = ap.parse_args()

pkg = _bootstrap.load()
from continual_skeletons_amd import parallel  # noqa: E402

dev = torch.device("cuda:0")


V = 18 if args.model == "coagcn" else 25


def make():
    if args.model == "coagcn":
        net = pkg.CoAGcn(pkg.kinetics_graph().A, bench.KIN_SHAPE, 400).eval()
        bench.randomise_(net, seed=0, attn_scale=1 / 18)
    elif args.model == "coagcn_ntu":
        net = pkg.CoAGcn(pkg.ntu_graph().A, (3, 300, 25, 2), 60).eval()
        bench.randomise_(net, seed=0, attn_scale=1 / 25)
    else:
        net = pkg.CoStGcn(pkg.ntu_graph().A).eval()
        bench.randomise_(net, seed=0)
    if args.force_ksplit >= 1:
        for blk in net.layers.values():
            if blk.out_channels >= args.ksplit_min_c:
                blk._pick_ksplit = lambda p, k=args.force_ksplit: k      # measurement only: overrides the block's policy
    if args.no_fuse:
        for blk in net.layers.values():
            blk.fuse_step = False
    if args.no_fuse_attention:
        for blk in net.layers.values():
            blk.gcn.fuse_embed_attention = False
    return net.to(dev)


eng = parallel.StreamShards(make, args.streams, args.shards, dev)
frames = torch.rand((8, args.streams, 3, V, 2), device=dev, generator=torch.Generator(device=dev).manual_seed(200))
warm = args.warm_cycles or (76 + 4 * 56 + args.fpl - 1) // args.fpl + 2
fi = 0


def cycle():
    global fi
    out = eng.forward_cycle([frames[(fi + f) % 8] for f in range(args.fpl)])
    fi += args.fpl
    return out


for _ in range(warm):
    cycle()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.cycles):
    out = cycle()
t_host = time.perf_counter() - t0          # launches enqueued (the host side of the timed region)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
assert out is not None and bool(torch.isfinite(out).all())
print(f"ONLINE_PASS model={args.model} fused={not args.no_fuse} shards={args.shards} fpl={args.fpl} streams={args.streams} warm_cycles={warm} cycles={args.cycles} "
      f"host_ms_per_cycle={t_host / args.cycles * 1e3:.4f} ms_per_cycle={dt / args.cycles * 1e3:.4f} frames_per_s={args.fpl * args.streams * args.cycles / dt:.0f}")
