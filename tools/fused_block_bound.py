#!/usr/bin/env python3
"""N1 (north_star: one fused kernel per clip block) for the 64-channel layers, priced by MEASUREMENT of its parts.

A clip block fused into one kernel keeps the post-GCN tensor y of a tile on chip; the 9 x 1 temporal conv of a tile of NT
positions needs y on NT + 8 V positions (4 frames of halo on either side), so every tile recomputes (NT + 8 V) / NT of its
graph conv.  At C = 64 the largest tile whose y (64 x (NT + 200) floats) fits a CU's LDS beside the operands is NT = 256
(117 KB): factor 456 / 256 = 1.78.  A fused kernel executes at least the MFMAs of
    graph conv on 1.78 x the frames  +  temporal conv,
at no better than the rates the stand-alone stage kernels reach on exactly that work (its y tile alone takes 117 KB of LDS:
ONE workgroup per CU, where the two-launch form runs 3 and 2).  This script measures both terms on the batch-256 L2-L4
shape and prints the bound beside the two-launch time and the HBM bytes either form moves.
usage: python tools/fused_block_bound.py [--batch 256]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import _bootstrap  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
a = ap.parse_args()
pkg = _bootstrap.load()
dev = "cuda:0"
A = pkg.ntu_graph().A
nm, V, T, C, NT = a.batch * 2, 25, 300, 64, 256
blk = pkg.SpatioTemporalBlock(C, C, A).eval().to(dev)
ops = blk._packed_ops(torch.device(dev))


def timed(fn, iters=6):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


x = torch.rand(nm, C, T, V, device=dev)
y = blk.gcn(x)
t_gcn = timed(lambda: blk.gcn(x))
t_tcn = timed(lambda: pkg.blocks.tcn_stage(y, ops["w"], ops["bias"], C, 9, 1, 4, relu=True, res_mode=1, x_res=x))
# the recomputed graph conv: the same kernel on (NT + 8 V) / NT times the frames (one tile's work scaled to the whole tensor)
factor = (NT + 8 * V) / NT
T_halo = int(round(T * factor))
xh = torch.rand(nm, C, T_halo, V, device=dev)
t_gcn_halo = timed(lambda: blk.gcn(xh))
tensor_gb = nm * C * T * V * 4 / 1e9
two_launch_gb = 5 * tensor_gb                      # x read (gcn), y write, y read, x read (residual), out write
fused_gb = (factor + 1) * tensor_gb                # x read with halo (the residual comes from the staged tile), out write
print(f"FUSED_BOUND batch={a.batch} L2-L4 shape (64->64, T=300, V=25): two launches {t_gcn:.3f} + {t_tcn:.3f} = {t_gcn + t_tcn:.3f} ms | "
      f"fused lower bound: graph conv on {factor:.2f}x the frames {t_gcn_halo:.3f} + temporal conv {t_tcn:.3f} = {t_gcn_halo + t_tcn:.3f} ms "
      f"({(t_gcn_halo + t_tcn) / (t_gcn + t_tcn) - 1:+.1%}) | HBM bytes per layer: two launches {two_launch_gb:.2f} GB, fused {fused_gb:.2f} GB "
      f"(-{two_launch_gb - fused_gb:.2f} GB)")
