#!/usr/bin/env python3
"""Few-stream latency of the online path: ms per frame-step and frames/s for 1..16 streams, default vs latency mode
(split-K in the TCN steps, CoStGcn.set_latency_mode)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _bootstrap, bench
pkg = _bootstrap.load()
dev = torch.device("cuda:0")
for streams in (1, 4, 16):
    frames = torch.rand((8, streams, 3, 25, 2), device=dev)
    for mode in ("default", "latency"):
        net = pkg.CoStGcn(pkg.ntu_graph().A).eval()
        bench.randomise_(net, seed=0)
        net = net.to(dev)
        if mode == "latency":
            net.set_latency_mode(8)
        for t in range(120):
            net.forward_step(frames[t % 8])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 400
        for t in range(n):
            net.forward_step(frames[t % 8])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        ks = [net.layers[f"layer{i + 1}"]._state.ksplit for i in range(10)]
        print(f"{streams:3d} stream(s) {mode:8s}: {dt * 1e3:.3f} ms/frame-step  {streams / dt:9.0f} frames/s   ksplit {ks}", flush=True)
