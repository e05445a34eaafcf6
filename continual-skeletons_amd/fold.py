"""Host-side weight folding / packing for the HIP stage kernels.

Eval-mode BatchNorm, conv biases and the adjacency re-weighting ``A * graph_attn``
(models/base.py:262) are constants of an inference module; they are folded ONCE per weight load into
the packed operands the kernels stream (layouts: include/cskel.h).  Folding is done in float64 and
rounded to fp32 once.
"""
from typing import Dict, Optional, Tuple

import numpy as np
import torch

KC = 16   # CSK_CPAD: packed weights zero-pad C_in to a multiple of this
MT = 64   # CSK_MT
BN_EPS = 1e-5


def _ceil_to(a: int, b: int) -> int:
    return (a + b - 1) // b * b


def bn_affine(weight, bias, mean, var, eps=BN_EPS) -> Tuple[torch.Tensor, torch.Tensor]:
    """Eval BatchNorm as y = x * s + t (float64)."""
    s = weight.double() / torch.sqrt(var.double() + eps)
    return s, bias.double() - mean.double() * s


def pack_conv_weight(weight: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """(C_out, C_in, K, 1) conv weight * per-output scale -> packed [K][C_in_pad][C_out_pad] fp32."""
    co, ci, k, _ = weight.shape
    w = weight.double()[:, :, :, 0] * scale[:, None, None]          # (co, ci, k)
    out = torch.zeros((k, _ceil_to(ci, KC), _ceil_to(co, MT)), dtype=torch.float64)
    out[:, :ci, :co] = w.permute(2, 1, 0)
    return out.float().contiguous()


SPLIT_KS = 16     # channels per K step of the bf16x3 split kernels (csrc/tcn_split.hip)


def split3_bf16(w32: torch.Tensor):
    """fp32 -> three bf16 pieces h + m + l (round to nearest even each; the subtractions are exact in fp32)."""
    h = w32.to(torch.bfloat16)
    r1 = w32 - h.float()
    m = r1.to(torch.bfloat16)
    l = (r1 - m.float()).to(torch.bfloat16)
    return h, m, l


def pack_conv_weight_split(weight: torch.Tensor, scale: torch.Tensor, stride: int = 1) -> torch.Tensor:
    """(C_out, C_in, K, 1) conv weight * per-output scale -> the operand image of the bf16x3 split kernels
    (include/cskel.h, csk_tcn_stage_bf16x3), an int16 tensor of bf16 bit patterns indexed
        [C_in_pad / 16][tap slots][3 pieces][2 channel halves][C_out_pad][8 channels]
    element [c16][slot][pc][h][co][j] = piece pc of W'[co, 16 c16 + 8 h + j, tap(slot)].  K = 9: nine slots in CLASS-MAJOR
    tap order -- taps 0, s, 2s, ... then 1, 1 + s, ... for stride s (the taps of a residue class read one de-interleaved
    set of source frames; for s = 1 the natural order).  K = 1 (the residual conv): three slots, tap 0 and two zero slots.
    The pieces are taken from the SAME fp32 value the exact-fp32 path packs (fp64 fold, one rounding to fp32)."""
    if weight.dim() == 4:
        weight = weight[:, :, :, 0]
    co, ci, k = weight.shape
    if k not in (1, 3, 9):
        raise ValueError(f"split weights are built for the 9 x 1 temporal conv, the 1 x 1 residual convs and the 3 graph-conv "
                         f"subsets, got k = {k}")
    w32 = (weight.double() * scale[:, None, None]).float()                         # (co, ci, k), as pack_conv_weight
    cpad, mpad = _ceil_to(ci, SPLIT_KS), _ceil_to(co, MT)
    taps = [r for rho in range(min(stride, k)) for r in range(rho, k, stride)] if k > 1 else [0]
    slots = _ceil_to(len(taps), 3)
    full = torch.zeros((mpad, cpad, slots), dtype=torch.float32)
    full[:co, :ci, : len(taps)] = w32[:, :, taps]
    pieces = torch.stack([p.view(torch.int16) for p in split3_bf16(full)], 0)        # (3, mpad, cpad, slots)
    out = pieces.view(3, mpad, cpad // SPLIT_KS, 2, 8, slots).permute(2, 5, 0, 3, 1, 4)  # [c16][slot][pc][h][co][j]
    return out.contiguous().reshape(-1)


def pad_vec(v: torch.Tensor) -> torch.Tensor:
    out = torch.zeros(_ceil_to(v.numel(), MT), dtype=torch.float64)
    out[: v.numel()] = v
    return out.float().contiguous()


def ell_from_dense(a_eff: torch.Tensor):
    """Column-wise ELL form of a (3, V, V) adjacency: for subset i and output joint w, the rows v with
    A[i, v, w] != 0 (the aggregation ``x @ A[i]`` sums over v).  Returns (src int32 [3,V,EW],
    val fp32 [3,V,EW], cnt int32 [3]); padding entries are (0, 0.0)."""
    a = a_eff.detach().cpu().double().numpy()
    three, v, _ = a.shape
    nz = a != 0
    per_col = nz.sum(axis=1)                      # (3, V): non-zeros per column
    cnt = per_col.max(axis=1).astype(np.int32)    # per subset
    ew = max(1, int(cnt.max()))
    src = np.zeros((three, v, ew), dtype=np.int32)
    val = np.zeros((three, v, ew), dtype=np.float32)
    for i in range(three):
        for w in range(v):
            rows = np.nonzero(nz[i, :, w])[0]
            src[i, w, : len(rows)] = rows
            val[i, w, : len(rows)] = a[i, rows, w]
    return torch.from_numpy(src), torch.from_numpy(val), torch.from_numpy(cnt), ew


def fold_graph_conv(sd: Dict[str, torch.Tensor], p: str = "", split: bool = False) -> dict:
    """Packed operands of csk_gcn_stage_f32 from a GraphConvolution state_dict (keys of
    models/base.py:231-258).  split: also the bf16x3 operand images of csk_gcn_stage_bf16x3 (the three subsets in the
    role of three taps; the conv gcn_residual as a one-tap image)."""
    sd = {k: v.detach().cpu() for k, v in sd.items() if k.startswith(p)}
    s, t = bn_affine(sd[p + "bn.weight"], sd[p + "bn.bias"], sd[p + "bn.running_mean"], sd[p + "bn.running_var"])
    co, ci = sd[p + "g_conv.0.weight"].shape[:2]
    conv_res = (p + "gcn_residual.0.weight") in sd
    r = 4 if conv_res else 3
    w = torch.zeros((r, _ceil_to(ci, KC), _ceil_to(co, MT)), dtype=torch.float64)
    bias = t.clone()
    for i in range(3):
        w[i, :ci, :co] = (sd[f"{p}g_conv.{i}.weight"].double()[:, :, 0, 0] * s[:, None]).t()
        bias += s * sd[f"{p}g_conv.{i}.bias"].double()
    if conv_res:
        sr, tr = bn_affine(sd[p + "gcn_residual.1.weight"], sd[p + "gcn_residual.1.bias"],
                           sd[p + "gcn_residual.1.running_mean"], sd[p + "gcn_residual.1.running_var"])
        w[3, :ci, :co] = (sd[p + "gcn_residual.0.weight"].double()[:, :, 0, 0] * sr[:, None]).t()
        bias += sr * sd[p + "gcn_residual.0.bias"].double() + tr
    a_eff = sd[p + "A"].double() * sd[p + "graph_attn"].double()       # models/base.py:262
    src, val, cnt, ew = ell_from_dense(a_eff)
    out = dict(w=w.float().contiguous(), bias=pad_vec(bias), ell_src=src, ell_val=val, ell_cnt_host=cnt,
               ell_w=ew, c_in=ci, c_out=co, res_mode=2 if conv_res else 1, V=a_eff.shape[-1], w_split=None, w_res_split=None)
    if split:
        one = torch.ones(co, dtype=torch.float64)                      # the BN scale is already folded into w
        w3 = w[:3, :ci, :co].permute(2, 1, 0).contiguous()             # (co, ci, 3 subsets), fp64
        out["w_split"] = pack_conv_weight_split(w3, one)
        if conv_res:
            out["w_res_split"] = pack_conv_weight_split(w[3, :ci, :co].t().contiguous().unsqueeze(-1), one)
    return out


def fold_temporal_conv(sd: Dict[str, torch.Tensor], p: str = "") -> dict:
    """Packed weight + bias of a TemporalConvolution (models/base.py:279-304)."""
    s, t = bn_affine(sd[p + "bn.weight"].cpu(), sd[p + "bn.bias"].cpu(), sd[p + "bn.running_mean"].cpu(),
                     sd[p + "bn.running_var"].cpu())
    wt = sd[p + "t_conv.weight"].detach().cpu()
    bias = s * sd[p + "t_conv.bias"].detach().cpu().double() + t
    return dict(w=pack_conv_weight(wt, s), bias=bias, c_in=wt.shape[1], c_out=wt.shape[0], k=wt.shape[2])


def fold_temporal_conv_split(sd: Dict[str, torch.Tensor], p: str = "", stride: int = 1) -> torch.Tensor:
    """bf16x3 split image of a TemporalConvolution's weight (same BN fold as fold_temporal_conv)."""
    s, _ = bn_affine(sd[p + "bn.weight"].cpu(), sd[p + "bn.bias"].cpu(), sd[p + "bn.running_mean"].cpu(),
                     sd[p + "bn.running_var"].cpu())
    return pack_conv_weight_split(sd[p + "t_conv.weight"].detach().cpu(), s, stride)


def fold_block_tail(sd: Dict[str, torch.Tensor], p: str = "", has_conv_residual: Optional[bool] = None,
                    split: bool = False, stride: int = 1) -> dict:
    """Operands of csk_tcn_stage_f32 for a whole SpatioTemporalBlock tail: tcn (+ conv residual).  split: also the
    bf16x3 operand images ``w_split`` / ``w_res_split`` (precision mode "bf16x3", 9-tap convs only)."""
    main = fold_temporal_conv(sd, p + "tcn.")
    out = dict(w=main["w"], k=main["k"], c=main["c_in"], c_out=main["c_out"], w_res=None, c_res=0, w_split=None, w_res_split=None)
    bias = main["bias"]
    if has_conv_residual is None:
        has_conv_residual = (p + "residual.t_conv.weight") in sd
    if has_conv_residual:
        res = fold_temporal_conv(sd, p + "residual.")
        out["w_res"], out["c_res"] = res["w"], res["c_in"]
        bias = bias + res["bias"]
    if split:
        out["w_split"] = fold_temporal_conv_split(sd, p + "tcn.", stride)
        if has_conv_residual:
            out["w_res_split"] = fold_temporal_conv_split(sd, p + "residual.")
    out["bias"] = pad_vec(bias)
    return out


def fold_data_bn(sd: Dict[str, torch.Tensor], p: str = "data_bn."):
    """BatchNorm1d over M*V*C channels (models/st_gcn/st_gcn.py:28,51) as per-channel scale/shift."""
    s, t = bn_affine(sd[p + "weight"].cpu(), sd[p + "bias"].cpu(), sd[p + "running_mean"].cpu(),
                     sd[p + "running_var"].cpu())
    return s.float().contiguous(), t.float().contiguous()
