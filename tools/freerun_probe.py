"""Does the per-cycle fork/join between the stream shards cost throughput?  Same engine, joined vs free-running
cycles (each shard enqueues on its own HIP stream without waiting for the others), interleaved in one process."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import _bootstrap
import bench

pkg = _bootstrap.load()
from continual_skeletons_amd import parallel

dev = torch.device("cuda:0")
streams, shards = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 2
fpl = int(sys.argv[2]) if len(sys.argv) > 2 else 4


def make():
    net = pkg.CoStGcn(pkg.ntu_graph().A).eval()
    bench.randomise_(net, seed=0)
    return net.to(dev)


eng = parallel.StreamShards(make, streams, shards, dev)
frames = torch.rand((8, streams, 3, 25, 2), device=dev)
for t in range(76 + 4 * 55):
    eng.forward_cycle([frames[t % 8]])
torch.cuda.synchronize()
slices = [[frames[i][lo:hi] for i in range(8)] for (lo, hi) in eng.bounds]


def joined(n):
    for c in range(n):
        eng.forward_cycle([frames[(c * fpl + f) % 8] for f in range(fpl)])


def free(n):
    for c in range(n):
        for k, (model, st) in enumerate(zip(eng.models, eng.streams)):
            with torch.cuda.stream(st):
                model.forward_cycle([slices[k][(c * fpl + f) % 8] for f in range(fpl)])


for rep in range(3):
    for name, fn in (("joined", joined), ("free", free)):
        n = 96 // fpl
        fn(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"rep {rep} {name:7s} shards {shards} fpl {fpl}: {fpl * streams * n / dt:,.0f} frames/s", flush=True)
