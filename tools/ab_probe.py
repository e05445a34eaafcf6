#!/usr/bin/env python3
"""Interleaved in-process A/B of a diagnostic env switch on the TCN stage (guide rule 24).
usage: python tools/ab_probe.py CSK_NOPRIO [BASE_VAR ...]  -- BASE_VARs are set in BOTH arms (e.g. CSK_TCN_NOLW)"""
import os, sys
os.environ["CSK_DIAG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, statistics
import _bootstrap
pkg = _bootstrap.load()
var, _, val = sys.argv[1].partition("=")
val = val or "1"
for b in sys.argv[2:]:
    os.environ[b] = "1"
dev = "cuda:0"
V = int(os.environ.get("AB_V", "25")); NM = int(os.environ.get("AB_NM", "512"))
A = (pkg.ntu_graph() if V == 25 else pkg.kinetics_graph()).A
for (ci, co, s, t) in [(64, 64, 1, 300), (64, 128, 2, 300), (128, 128, 1, 150), (128, 256, 2, 150), (256, 256, 1, 75)]:
    blk = pkg.SpatioTemporalBlock(ci, co, A, stride=s).eval().to(dev)
    x = torch.rand(NM, ci, t, V, device=dev); y = blk.gcn(x); ops = blk._packed_ops(x.device)
    conv = ci != co or s != 1
    res = {0: [], 1: []}
    for rnd in range(12):
        for flag in (0, 1):
            if flag: os.environ[var] = val
            else: os.environ.pop(var, None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = pkg.blocks.tcn_stage(y, ops["w"], ops["bias"], co, 9, s, 4, relu=True, res_mode=2 if conv else 1, x_res=x,
                                       w_res=ops.get("w_res") if conv else None)
            e1.record(); torch.cuda.synchronize()
            if rnd >= 2: res[flag].append(e0.elapsed_time(e1))
    os.environ.pop(var, None)
    m0, m1 = statistics.median(res[0]), statistics.median(res[1])
    print(f"{ci}->{co} s{s}: default {m0:.3f} ms (min {min(res[0]):.3f}) | {var}={val} {m1:.3f} ms (min {min(res[1]):.3f}) | ratio {m1/m0:.4f}")
