#!/usr/bin/env python3
"""In-process A/B of the TCN stage per layer shape (batch 256, NTU): exact fp32 (csk_tcn_stage_f32) vs the opt-in bf16x3
split kernel (csk_tcn_stage_bf16x3), interleaved rounds, median ms, fp32-equivalent TFLOP/s, max |difference|.
usage: python tools/ab_split_probe.py [batch] [rounds]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import _bootstrap  # noqa: E402
import bench  # noqa: E402

pkg = _bootstrap.load()
dev = "cuda:0"
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
A = pkg.ntu_graph().A
shapes = [(64, 64, 1, 300), (64, 128, 2, 300), (128, 128, 1, 150), (128, 256, 2, 150), (256, 256, 1, 75)]
for ci, co, s, T in shapes:
    blk = pkg.SpatioTemporalBlock(ci, co, A, stride=s).eval()
    bench.randomise_(blk, 0)
    blk = blk.to(dev)
    x = torch.rand((2 * batch, ci, T, 25), device=dev)
    y = blk.gcn(x)
    ops32 = blk._packed_ops(dev)
    pkg.set_precision(blk, "bf16x3")
    ops3 = blk._packed_ops(dev)
    mode = 1 if blk.residual is pkg.unity else 2
    t_out = (T - 1) // s + 1
    out = torch.empty((2 * batch, co, t_out, 25), device=dev)

    def run(split):
        o = ops3 if split else ops32
        return pkg.blocks.tcn_stage(y, o["w_split"] if split else o["w"], o["bias"], co, 9, s, 4, relu=True, res_mode=mode, x_res=x,
                                    w_res=o["w_res_split"] if split else o["w_res"], out=out, split=split)
    ref = run(False).clone()
    got = run(True).clone()
    err = float((ref - got).abs().max())
    times = {False: [], True: []}
    for _ in range(rounds):
        for split in (False, True):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run(split)
            e1.record()
            torch.cuda.synchronize()
            times[split].append(e0.elapsed_time(e1) / 3)
    flop = 2.0 * 2 * batch * t_out * 25 * (9 * co * co + (ci * co if mode == 2 else 0))
    m32, m3 = statistics.median(times[False]), statistics.median(times[True])
    if os.environ.get("CSK_DIAG"):          # where does the split kernel's time go?  (results are garbage with skips on)
        parts = []
        for bits in (0, 16, 0, 16):
            os.environ["CSK_SPLIT_SKIP"] = str(bits)
            run(True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run(True)
            e1.record()
            torch.cuda.synchronize()
            parts.append(f"skip{bits}={e0.elapsed_time(e1) / 3:.3f}")
        os.environ["CSK_SPLIT_SKIP"] = "0"
        print("   SPLIT_DIAG (0 = staggered wave halves, 16 = all waves stage first): " + " ".join(parts))
    print(f"SPLIT_AB {ci}->{co} s{s} T={T}: f32 {m32:.3f} ms ({flop / m32 / 1e9:.1f} TF)  bf16x3 {m3:.3f} ms ({flop / m3 / 1e9:.1f} TF-equiv)  "
          f"speedup {m32 / m3:.2f}x  max|diff| {err:.2e}  |out|max {float(ref.abs().max()):.2f}")
