#!/usr/bin/env python3
"""Diagnostic: per-workgroup s_memtime stamps of tcn_stage_kernel (prologue / K loop / epilogue split)."""
import os, sys
os.environ["CSK_DIAG"] = "1"   # must be set before the library is loaded
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import _bootstrap
pkg = _bootstrap.load()
from continual_skeletons_amd.models import layer_table
dev = "cuda:0"
A = pkg.ntu_graph().A
for (ci, co, s, res, t) in [(64, 64, 1, True, 300), (128, 128, 1, True, 150), (256, 256, 1, True, 75)]:
    nm = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    blk = pkg.SpatioTemporalBlock(ci, co, A, stride=s, residual=res).eval().to(dev)
    x = torch.rand(nm, ci, t, 25, device=dev)
    y = blk.gcn(x)
    ops = blk._packed_ops(x.device)
    NT = 128 if co % 128 == 0 else 256
    nwg = ((t * 25 + NT - 1) // NT) * nm * max(1, co // 128 if co % 128 == 0 else co // 64)
    stamps = torch.zeros(nwg * 6 + nwg * 16, dtype=torch.int64, device=dev)
    for it in range(2):
        os.environ["CSK_STAMPS"] = str(stamps.data_ptr()) if it == 1 else "0"
        if it == 0: os.environ.pop("CSK_STAMPS")
        out = pkg.blocks.tcn_stage(y, ops["w"], ops["bias"], ops["c_out"], 9, s, 4, relu=True, res_mode=1, x_res=x, w_res=None)
        torch.cuda.synchronize()
    os.environ.pop("CSK_STAMPS", None)
    allst = stamps.cpu().numpy()
    st = allst[: nwg * 6].reshape(nwg, 6)
    ph = allst[nwg * 6:].reshape(nwg, 4, 4)
    chunks = max(1, co // 8 - 1)
    print(f"   per-chunk per-wave cycles: barrier1-wait {np.median(ph[:,:,0])/chunks:.0f}  commit+barrier2 {np.median(ph[:,:,1])/chunks:.0f}  issue {np.median(ph[:,:,2])/chunks:.0f}  mfma {np.median(ph[:,:,3])/chunks:.0f}  (ideal mfma alone 9216)")
    pro, loop, epi = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2]
    tot = st[:, 3] - st[:, 0]
    print(f"C={co}: WGs {nwg}; cycles (s_memtime ticks = 100MHz? check): prologue med {np.median(pro):.0f} p90 {np.percentile(pro,90):.0f} | loop med {np.median(loop):.0f} | epilogue med {np.median(epi):.0f} p90 {np.percentile(epi,90):.0f} | total med {np.median(tot):.0f}")
    span = st[:, 3].max() - st[:, 0].min()
    print(f"   kernel span {span} ticks; sum(total)/span = {tot.sum()/span:.1f} concurrent WGs; start spread first 512: {np.sort(st[:,0])[min(511, len(st) - 1)]-st[:,0].min()}")
    # gap between a WG end and the next WG start on the same CU slot
    hw = st[:, 4] & 0xFFFFFFFF
    cu_key = (st[:, 5] << 32) | (hw & 0xFFFF00)   # xcc + se/sh/cu bits (approx)
    gaps = []
    for key in np.unique(cu_key)[:64]:
        sel = st[cu_key == key]
        ends = np.sort(sel[:, 3]); starts = np.sort(sel[:, 0])
        if len(sel) > 4:
            gaps.append(len(sel))
    print("   WGs per (xcc,cu-ish) key sample:", gaps[:8])
