#!/usr/bin/env python3
"""Diagnostic: per-wave s_memtime stamps of tcn_step16_kernel at the online shapes (1024 NTU streams): where a chunk's cycles go
(wait at the first barrier / commit + second barrier / MFMA segments with the next chunk's loads) and prologue / epilogue."""
import os, sys
os.environ["CSK_DIAG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import _bootstrap
pkg = _bootstrap.load()
from continual_skeletons_amd import native
dev = torch.device("cuda:0")

def pair_report(ws, we, hwid, nwg):
    """who shares a CU: s_memtime has a different base per XCD (the start stamps fall into clusters); HW_ID bits 8-15 = CU / SH / SE"""
    order = np.argsort(ws); gaps = np.diff(ws[order]); cuts = np.where(gaps > 10_000_000)[0]
    xcd = np.zeros(nwg, dtype=int); xcd[order] = np.searchsorted(cuts, np.arange(nwg), side="left")
    cu = (hwid >> 8) & 0xFF
    dur = we - ws
    pairs = {}
    for w in range(nwg):
        pairs.setdefault((int(xcd[w]), int(cu[w])), []).append(w)
    two = [v for v in pairs.values() if len(v) == 2]
    if two:
        d = np.array([[dur[a], dur[b]] for a, b in two]); st_d = np.array([[ws[a], ws[b]] for a, b in two])
        first_is_fast = np.mean((st_d[:, 0] < st_d[:, 1]) == (d[:, 0] < d[:, 1]))
        print(f"STAMP16   {len(pairs)} CUs seen, {len(two)} hold two workgroups: |duration difference| inside a CU median {np.median(np.abs(d[:, 0] - d[:, 1])):.0f}, "
              f"faster one median {np.median(d.min(axis=1)):.0f}, slower one {np.median(d.max(axis=1)):.0f}; the one that started first is the faster one in {first_is_fast:.2f} of the CUs; "
              f"start offset inside a CU median {np.median(np.abs(st_d[:, 0] - st_d[:, 1])):.0f}; per-CU max duration p1 / p50 / p99 {np.percentile(d.max(axis=1), [1, 50, 99]).round()}")

P = 1024 * 2 * 25
lib = native.lib()
for (c, n_emit, hs, res) in [(64, 4, 1, 1), (64, 4, 1, 0), (128, 2, 1, 1), (256, 1, 1, 1)]:
    tc = pkg.TemporalConvolution(c, c, kernel_size=9, stride=1, padding=4).eval().to(dev)
    ops = tc._packed_ops(dev)
    slots = 16
    ring = torch.rand((slots, c, P), device=dev)
    xres = torch.rand((8, c, P), device=dev)
    out = torch.empty((4, c, P), device=dev)
    nwg = 512
    stamps = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=dev)
    for it in range(3):
        if it == 2:
            os.environ["CSK_STAMPS"] = str(stamps.data_ptr())
        rc = lib.csk_tcn_step_f32(native.ptr(ring), slots, 11, hs, n_emit, native.ptr(ops["w"]), native.ptr(xres) if res else None, 8, 0, 1,
                                  None, native.ptr(ops["bias"]), native.ptr(out), 4, 0, c, c, P, 9, res, c if res else 0, 1, 1, None,
                                  native.stream_of(ring))
        native.check(rc, "csk_tcn_step_f32")
        torch.cuda.synchronize()
    os.environ.pop("CSK_STAMPS", None)
    st = stamps.cpu().numpy().reshape(nwg, 4, 8)
    chunks = c // 4
    pro, loop, epi = st[:, :, 1] - st[:, :, 0], st[:, :, 2] - st[:, :, 1], st[:, :, 3] - st[:, :, 2]
    span = st[:, :, 3].max() - st[:, :, 0].min()
    print(f"STAMP16 C={c} n_emit={n_emit} res={res}: span {span} ticks | per wave: prologue {np.median(pro):.0f} loop {np.median(loop):.0f} epilogue {np.median(epi):.0f} "
          f"| per chunk: barrier1 wait {np.median(st[:, :, 4]) / chunks:.0f}  commit+barrier2 {np.median(st[:, :, 5]) / chunks:.0f}  mfma+issue {np.median(st[:, :, 6]) / chunks:.0f} "
          f"(225 MFMAs = 7200 cycles alone, 14400 shared) | start spread {np.percentile(st[:, :, 0], 99) - st[:, :, 0].min():.0f} | end spread {st[:, :, 3].max() - np.percentile(st[:, :, 3], 1):.0f}")
    # per workgroup: start (first wave in) and end (last wave out) relative to the launch's first stamp, and its duration
    t0 = st[:, :, 0].min()
    ws, we = st[:, :, 0].min(axis=1) - t0, st[:, :, 3].max(axis=1) - t0
    pc = lambda a: " / ".join(f"{np.percentile(a, q):.0f}" for q in (1, 25, 50, 75, 99, 100))
    print(f"STAMP16   workgroups (p1 / p25 / p50 / p75 / p99 / max): start {pc(ws)} | end {pc(we)} | duration {pc(we - ws)}")
    pair_report(ws, we, st[:, 0, 7], nwg)
# graph conv (gcn16_kernel): phase sums per wave and chunk of 8 channels
A = pkg.ntu_graph().A
for (ci, co, frames) in [(64, 64, 4), (128, 128, 2), (256, 256, 1)]:
    g = pkg.GraphConvolution(ci, co, A).eval().to(dev)
    x = torch.rand((frames, ci, P), device=dev)
    y = torch.empty((frames, co, P), device=dev)
    nwg = 512
    stamps = torch.zeros(nwg * 4 * 10, dtype=torch.int64, device=dev)
    for it in range(3):
        if it == 2:
            os.environ["CSK_STAMPS"] = str(stamps.data_ptr())
        g.stage(x, y, n_seg=frames, frames=2048, x_strides=(ci * P, P), y_strides=(co * P, P))
        torch.cuda.synchronize()
    os.environ.pop("CSK_STAMPS", None)
    raw = stamps.cpu().numpy()
    st = raw[: nwg * 32].reshape(nwg, 4, 8)
    se = raw[nwg * 32:].reshape(nwg, 4, 2)
    print(f"STAMP16 gcn {ci}->{co} x{frames}: per wave (median): prologue {np.median(st[:, :, 0] - se[:, :, 0]):.0f}  K loop {np.median(st[:, :, 1] - st[:, :, 0]):.0f}  epilogue (stores retired) "
          f"{np.median(se[:, :, 1] - st[:, :, 1]):.0f} | workgroup duration p1 / p50 / p99 / max {np.percentile(se[:, :, 1].max(axis=1) - se[:, :, 0].min(axis=1), [1, 50, 99, 100]).round()}")
    pair_report(se[:, :, 0].min(axis=1), se[:, :, 1].max(axis=1), st[:, 0, 7], nwg)
    chunks = max(ci, 16) // 8
    hw = st[:, 0, 7]
    print(f"STAMP16 gcn {ci}->{co} x{frames}: loop {np.median(st[:, :, 1] - st[:, :, 0]):.0f} cycles = {chunks} chunks | per chunk and wave: aggregation+W commit {np.median(st[:, :, 2]) / chunks:.0f}  "
          f"barrier1 wait {np.median(st[:, :, 3]) / chunks:.0f}  x commit+issue {np.median(st[:, :, 4]) / chunks:.0f}  mfma {np.median(st[:, :, 5]) / chunks:.0f} (150 MFMAs = 4800 alone, 9600 shared)  "
          f"barrier2 wait {np.median(st[:, :, 6]) / chunks:.0f} | wave-slot parity of wave 0: {np.bincount((hw & 1).astype(int), minlength=2)} | start offset between odd and even slots "
          f"{np.median(st[(hw & 1) == 1, 0, 0]) - np.median(st[(hw & 1) == 0, 0, 0]) if ((hw & 1) == 1).any() and ((hw & 1) == 0).any() else float('nan'):.0f}")
