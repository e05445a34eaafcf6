#!/usr/bin/env python3
"""In-process A/B of the graph-conv stage per layer shape (batch 256, NTU): exact fp32 (csk_gcn_stage_f32) vs the opt-in
bf16x3 channel mix (csk_gcn_stage_bf16x3), interleaved rounds, median ms, max |difference|.
usage: python tools/ab_gcn_split_probe.py [batch] [rounds]"""
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
for ci, co, T in [(64, 128, 300), (128, 128, 150), (128, 256, 150), (256, 256, 75)]:
    g32 = pkg.GraphConvolution(ci, co, A).eval()
    bench.randomise_(g32, 0)
    g3 = pkg.GraphConvolution(ci, co, A).eval()
    g3.load_state_dict(g32.state_dict())
    g3.precision = "bf16x3"
    g32, g3 = g32.to(dev), g3.to(dev)
    x = torch.rand((2 * batch, ci, T, 25), device=dev)
    ref, got = g32(x), g3(x)
    err = float((ref - got).abs().max())
    times = {0: [], 1: []}
    for _ in range(rounds):
        for k, m in enumerate((g32, g3)):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                m(x)
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / 3)
    m32, m3 = statistics.median(times[0]), statistics.median(times[1])
    print(f"GCN_SPLIT_AB {ci}->{co} T={T}: f32 {m32:.3f} ms  bf16x3 {m3:.3f} ms  speedup {m32 / m3:.2f}x  max|diff| {err:.2e}  |out|max {float(ref.abs().max()):.2f}")
