"""In-process repeatability probe of the online workload: same box, same clocks, repeated engine construction.
usage: python tools/fpl_probe.py MODE   (MODE = keepmem | freemem)
keepmem: the caching allocator hands every new engine the same blocks (only the HIP streams differ);
freemem: blocks are returned to the driver between engines (new physical placement, new streams)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import _bootstrap
import bench

pkg = _bootstrap.load()
from continual_skeletons_amd import parallel

dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "keepmem"
for rep in range(8 if mode != "map" else 0):
    fpl, shards, streams = 4, 2, 1024
    cycles = 96 // fpl
    dt, _, _, _ = bench.run_step_workload(pkg, dev, streams, cycles, 2, 0, 1, parallel, None, shards, fpl=fpl)
    print(f"{mode} rep {rep}: {fpl * streams * cycles / dt:,.0f} frames/s  reserved {torch.cuda.memory_reserved() / 1e9:.1f} GB", flush=True)
    if mode == "freemem":
        torch.cuda.empty_cache()

if mode == "map":       # which pool streams overlap with the current stream / each other
    cur = torch.cuda.current_stream(dev)
    ss = [torch.cuda.Stream(device=dev) for _ in range(10)]
    print("vs current:", [int(parallel.streams_overlap(s, cur)) for s in ss])
    for i, a in enumerate(ss):
        print(i, [int(parallel.streams_overlap(a, b)) if a is not b else "-" for b in ss])
