#!/usr/bin/env python3
"""Why is the GCN stage slower on the continual path's channel-major frames than on clips?  Runs csk_gcn_stage_f32 on
step-shaped operands -- n_seg ring slots of (C, P) frames, P = 51200 positions (1024 streams) -- for n_seg = 1 .. 16, so
that the launch grows from 800 to 12800 workgroups at unchanged access pattern, next to the clip-shaped launch of the
same layer (512 sequences).  If the rate climbs to the clip rate with n_seg, the online loss is tile quantisation /
launch tails; if it stays low, it is the access pattern of the state layout."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import _bootstrap  # noqa: E402
import bench  # noqa: E402

pkg = _bootstrap.load()
dev = "cuda:0"
A = pkg.ntu_graph().A
P, n_skel = 51200, 2048


def timed(fn, it=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


for (ci, co, t_clip) in [(64, 64, 300), (128, 128, 150), (256, 256, 75)]:
    g = pkg.GraphConvolution(ci, co, A).eval()
    bench.randomise_(g, 0)
    g = g.to(dev)
    r = 3
    x = torch.rand(512, ci, t_clip, 25, device=dev)
    ms = timed(lambda: g(x))
    fl = 2.0 * (r * ci * co + 6 * ci) * t_clip * 25 * 512
    print(f"C {ci}->{co}: clip-shaped (512 seq x {t_clip} frames)      {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TFLOP/s executed")
    for n_seg in (1, 2, 4, 8, 16):
        xs = torch.rand(n_seg, ci, P, device=dev)
        ys = torch.empty(n_seg, co, P, device=dev)
        ms = timed(lambda: g.stage(xs, ys, n_seg=n_seg, frames=n_skel, x_strides=(ci * P, P), y_strides=(co * P, P)))
        fl = 2.0 * (r * ci * co + 6 * ci) * n_skel * 25 * n_seg
        print(f"           step-shaped, {n_seg:2d} frame slots of (C, {P})  {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TFLOP/s executed")
