#!/usr/bin/env python3
"""Race screen for the fused continual block kernel (its phase-G -> phase-T hand-off goes through L2 inside one
workgroup): 1024 streams, fused against two-launch engines fed the same frames for many cycles; every prediction and,
at the end, every state ring must be bit-identical.  usage: python tools/fused_stress.py [cycles]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import _bootstrap  # noqa: E402
import bench  # noqa: E402

pkg = _bootstrap.load()
dev = torch.device("cuda:0")
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 150
streams = 1024


def make(fuse):
    net = pkg.CoStGcn(pkg.ntu_graph().A, pool_size=3, pool_padding=1).eval()
    bench.randomise_(net, 0)
    for blk in net.layers.values():
        blk.fuse_step = fuse
    return net.to(dev)


a, b = make(True), make(False)
frames = torch.rand((16, streams, 3, 25, 2), device=dev)
bad = n = 0
for c in range(cycles):
    fr = [frames[(4 * c + f) % 16] * (1.0 + 0.01 * (c % 7)) for f in range(4)]
    oa, ob = a.forward_cycle(fr), b.forward_cycle(fr)
    assert len(oa) == len(ob)
    for x, y in zip(oa, ob):
        n += 1
        bad += int(not torch.equal(x, y))
rings = 0
for i in range(1, 11):
    sa, sb = a.layers[f"layer{i}"]._state, b.layers[f"layer{i}"]._state
    rings += int(not (torch.equal(sa.y, sb.y) and torch.equal(sa.out, sb.out)))
print(f"fused vs two-launch, {streams} streams, {cycles} cycles: {n} predictions compared, {bad} differ; {rings} of 10 layers' rings differ")
sys.exit(1 if (bad or rings or n == 0) else 0)
