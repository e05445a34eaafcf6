#!/usr/bin/env python3
"""rocprofv3 target: ST-GCN clip forwards (bench.py's clip shape: batch 256, NTU-60) in one precision mode.
usage: python tools/clip_pass.py [--precision f32|bf16x3] [--batch 256] [--forwards 5]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import _bootstrap  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="f32", choices=["f32", "bf16x3"])
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--forwards", type=int, default=5)
a = ap.parse_args()
pkg = _bootstrap.load()
dev = "cuda:0"
net = pkg.StGcn(pkg.ntu_graph().A).eval()
bench.randomise_(net, 0)
net = net.to(dev)
if a.precision != "f32":
    pkg.set_precision(net, a.precision)
x = torch.rand((a.batch, 3, 300, 25, 2), device=dev, generator=torch.Generator(device=dev).manual_seed(100))
for _ in range(2):
    net(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.forwards):
    out = net(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.forwards
assert bool(torch.isfinite(out).all())
print(f"CLIP_PASS precision={a.precision} batch={a.batch} forwards={a.forwards} ms_per_forward={dt * 1e3:.3f} clips_per_s={a.batch / dt:.1f}")
