"""Interleaved in-process A/B of a diagnostic env switch on the A-GCN clip forward (Kinetics shape, batch 64).
usage: python tools/ab_agcn_probe.py CSK_SLOW_EPI | CSK_TCN16=2"""
import os, sys, time, statistics
VAR = sys.argv[1] if len(sys.argv) > 1 else "CSK_SLOW_EPI"
VAR, VAL = VAR.split("=") if "=" in VAR else (VAR, "1")
os.environ["CSK_DIAG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, _bootstrap, bench
pkg = _bootstrap.load()
dev = "cuda:0"; A = pkg.kinetics_graph().A; shape = (3, 300, 18, 2)
x = torch.rand((64,) + shape, device=dev)
net = pkg.AGcn(A, shape, 400).eval(); bench.randomise_(net, 0); net = net.to(dev)
res = {0: [], 1: []}
for rnd in range(8):
    for flag in (0, 1):
        if flag: os.environ[VAR] = VAL
        else: os.environ.pop(VAR, None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): net(x)
        torch.cuda.synchronize()
        if rnd >= 2: res[flag].append((time.perf_counter() - t0) / 3 * 1e3)
print(f"AGCN clip b64: default {statistics.median(res[0]):.3f} ms | {VAR}={VAL} {statistics.median(res[1]):.3f} ms")
