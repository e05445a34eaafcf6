#!/usr/bin/env python3
"""Numerical study (CPU, oracle): error of split-bf16 emulation of the fp32 convs of the ST-GCN path.
x ~ x_hi + x_lo (+ x_lo2), each bf16; products of bf16 values are exact in fp32, so running the convs on the
rounded pieces in fp32 emulates bf16 MFMA with fp32 accumulation.  Schemes: 3 products (hh, hl, lh) and
6 products (3-way split: hh, hm, mh, hl, lh, mm)."""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests", "golden"))
import torch, torch.nn.functional as F
_conv2d = F.conv2d       # the real one (the oracle's F.conv2d is patched below)
from oracle import stgcn_oracle as o
from tests.helpers import g6_state_dict

def split(x, n):
    parts, r = [], x
    for _ in range(n):
        h = r.to(torch.bfloat16).to(torch.float32)
        parts.append(h); r = r - h
    return parts

def conv_split(x, w, scheme, **kw):
    if scheme == "fp32":
        return _conv2d(x, w, None, **kw)
    n = 2 if scheme == "x3" else 3
    xs, ws = split(x, n), split(w, n)
    pairs = [(0, 0), (0, 1), (1, 0)] if scheme == "x3" else [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]
    out = 0
    for i, j in reversed(pairs):          # small terms first
        out = out + _conv2d(xs[i], ws[j], None, **kw)
    return out

def run(scheme):
    orig = F.conv2d
    def patched(x, w, b=None, **kw):
        y = conv_split(x, w, scheme, **kw)
        return y if b is None else y + b.view(1, -1, 1, 1)
    o.F.conv2d = patched
    try:
        a, sd, x = g6_state_dict("ntu")
        taps = {}
        with torch.no_grad():
            lg = o.stgcn_forward(x[:1], sd, taps=taps)
        return lg, taps
    finally:
        o.F.conv2d = orig

ref, rt = run("fp32")
for sch in ("x3", "x6"):
    lg, t = run(sch)
    errs = {k: float((t[k] - rt[k]).abs().max()) for k in ("layer1", "layer5", "layer10")}
    print(f"{sch}: max|logit err| = {float((lg - ref).abs().max()):.2e}   activation errs {errs}   (|act| <= {float(rt['layer10'].abs().max()):.1f})")
