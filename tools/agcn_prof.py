#!/usr/bin/env python3
"""rocprofv3 target: A-GCN clip forwards at the Kinetics-400 shape (BASELINE configs[3], batch 64), same weights and
input as bench.py's agcn_kinetics leg; with "ntu" as third argument the NTU-60 shape (V = 25, 60 classes) instead.
usage: python tools/agcn_prof.py [batch] [forwards] [ntu]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import _bootstrap  # noqa: E402
import bench  # noqa: E402

pkg = _bootstrap.load()
dev = "cuda:0"
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ntu = len(sys.argv) > 3 and sys.argv[3] == "ntu"
shape = (3, 300, 25, 2) if ntu else bench.KIN_SHAPE
net = (pkg.AGcn(pkg.ntu_graph().A, shape, 60) if ntu else pkg.AGcn(pkg.kinetics_graph().A, shape, 400)).eval()
bench.randomise_(net, 0, attn_scale=1 / shape[2])
net = net.to(dev)
x = torch.rand((batch,) + shape, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
for _ in range(2):
    net(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    out = net(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
assert bool(torch.isfinite(out).all())
print(f"AGCN_PASS shape={'ntu' if ntu else 'kinetics'} batch={batch} forwards={n} ms_per_forward={dt * 1e3:.3f} clips_per_s={batch / dt:.1f}")
