#!/usr/bin/env python3
"""Diagnostic: per-wave s_memtime stamps of gcn_stage_sparse_kernel (prologue / K loop phases / last chunk / epilogue).
s_memtime ticks at 100 MHz on gfx950 (10 ns)."""
import os, sys
os.environ["CSK_DIAG"] = "1"   # must be set before the library is loaded
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import _bootstrap
pkg = _bootstrap.load()
dev = "cuda:0"
A = pkg.ntu_graph().A
NMS = [int(a) for a in sys.argv[1:]] or [512]
for (ci, co, t, nm) in [(c[0], c[1], c[2], nm) for nm in NMS for c in [(64, 64, 300), (64, 128, 300), (128, 128, 150), (256, 256, 75)]]:
    blk = pkg.SpatioTemporalBlock(ci, co, A, stride=1, residual=True).eval().to(dev)
    x = torch.rand(nm, ci, t, 25, device=dev)
    NT = 128 if co % 128 == 0 else 256
    nwg = ((t * 25 + NT - 1) // NT) * nm * max(1, co // 128 if co % 128 == 0 else co // 64)
    stamps = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(3):
        if it == 2:
            os.environ["CSK_STAMPS"] = str(stamps.data_ptr())
        ev0.record()
        y = blk.gcn(x)
        ev1.record()
        torch.cuda.synchronize()
        if it == 1:
            ms_plain = ev0.elapsed_time(ev1)
    ms_st = ev0.elapsed_time(ev1)
    os.environ.pop("CSK_STAMPS", None)
    st = stamps.cpu().numpy().reshape(nwg, 4, 8).astype(np.float64)
    chunks = max(1, (ci + 15) // 16 - 1)
    pro, loop, last, epi = st[..., 1] - st[..., 0], st[..., 2] - st[..., 1], st[..., 3] - st[..., 2], st[..., 4] - st[..., 3]
    tot = st[..., 4] - st[..., 0]
    med = lambda a: float(np.median(a))  # noqa: E731
    print(f"C {ci}->{co} nm={nm}: {nwg} WGs, {ms_plain:.3f} ms plain / {ms_st:.3f} ms stamped; ticks of 10 ns per wave (median):")
    print(f"   prologue {med(pro):.0f} | loop {med(loop):.0f} ({chunks} chunks: wait-barrier1 {med(st[..., 5]) / chunks:.0f} commit+barrier2 {med(st[..., 6]) / chunks:.0f}"
          f" issue+mfma {med(st[..., 7]) / chunks:.0f} per chunk) | last chunk incl. commit {med(last):.0f} | epilogue {med(epi):.0f} | total {med(tot):.0f}")
    span = st[..., 4].max() - st[..., 0].min()
    print(f"   kernel span {span:.0f} ticks; mean concurrent WGs {tot.mean(axis=1).sum() / span:.0f}")
