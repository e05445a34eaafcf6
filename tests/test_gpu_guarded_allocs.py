"""Whole models with every device buffer the package allocates sitting between NaN guards (tests/helpers.guarded_allocs): the
outputs must be bit for bit those of the plain run.  One pass over every kernel family -- default clip path, clip latency mode
(split-K stage kernels + reductions), the opt-in bf16x3 stages, A-GCN, the continual engines (Python and native plan, default and
latency mode) -- for reads outside an operand (round 6 found one in csk_tcn_step_f32 at test-only channel counts; on boxes where
the operand ended a mapping it was a GPU memory fault that took the whole pytest process down)."""
import pytest
import torch

import _bootstrap
from tests.helpers import guarded_allocs

pytestmark = pytest.mark.gpu
pkg = _bootstrap.load()
DEV = "cuda:0"


def _stgcn(seed=0):
    import bench
    net = pkg.StGcn(pkg.ntu_graph().A).eval()
    bench.randomise_(net, seed)
    return net.to(DEV)


@pytest.mark.parametrize("mode", ["default", "latency", "bf16x3"])
def test_stgcn_clip_forward_with_guarded_buffers(mode):
    x = torch.rand((3, 3, 52, 25, 2), generator=torch.Generator().manual_seed(1)).to(DEV)
    outs = []
    for guard in (False, True):
        net = _stgcn()
        if mode == "latency":
            net.set_latency_mode(4)
        if mode == "bf16x3":
            pkg.set_precision(net, "bf16x3")
        if guard:
            with guarded_allocs() as ga:
                outs.append(net(x).cpu())
            assert ga.count >= 20, ga.count          # every stage output (and scratch) went through the hook
        else:
            outs.append(net(x).cpu())
    assert bool(torch.isfinite(outs[1]).all()), "a kernel read outside a buffer"
    assert torch.equal(outs[0], outs[1])


def test_agcn_clip_forward_with_guarded_buffers():
    import bench
    x = torch.rand((2, 3, 40, 18, 2), generator=torch.Generator().manual_seed(2)).to(DEV)
    outs = []
    for guard in (False, True):
        net = pkg.AGcn(pkg.kinetics_graph().A, (3, 40, 18, 2), 400).eval()
        bench.randomise_(net, 0, attn_scale=1 / 18)
        net = net.to(DEV)
        if guard:
            with guarded_allocs():
                outs.append(net(x).cpu())
        else:
            outs.append(net(x).cpu())
    assert bool(torch.isfinite(outs[1]).all()), "a kernel read outside a buffer"
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("model,native_plan,latency", [("costgcn", True, False), ("costgcn", False, False), ("costgcn", True, True),
                                                      ("costgcn", False, True), ("coagcn", True, False), ("coagcn", False, False)])
def test_continual_stepping_with_guarded_buffers(model, native_plan, latency):
    """7 streams (a ragged tile), 104 frames: single frames, then 4-frame cycles; slab, scratch and every per-call buffer guarded."""
    import bench
    v = 25 if model == "costgcn" else 18
    frames = torch.rand((104, 7, 3, v, 2), generator=torch.Generator().manual_seed(3)).to(DEV)
    outs = []
    for guard in (False, True):
        if model == "costgcn":
            net = pkg.CoStGcn(pkg.ntu_graph().A, pool_size=2, pool_padding=0).eval()
            bench.randomise_(net, 0)
        else:
            net = pkg.CoAGcn(pkg.kinetics_graph().A, (3, 300, 18, 2), 400, pool_size=2, pool_padding=0).eval()
            bench.randomise_(net, 0, attn_scale=1 / 18)
        net = net.to(DEV)
        net.use_native_plan = native_plan
        if latency:
            net.set_latency_mode(2)

        def run():
            got = []
            for t in range(8):
                o = net.forward_step(frames[t])
                if o is not None:
                    got.append(o.cpu())
            for c in range(24):
                for o in net.forward_cycle([frames[8 + 4 * c + f] for f in range(4)]):
                    got.append(o.cpu())
            return got

        if guard:
            with guarded_allocs() as ga:
                outs.append(run())
            assert ga.count >= 3, ga.count           # the state slab, the pooling ring, per-cycle buffers
        else:
            outs.append(run())
    assert len(outs[0]) == len(outs[1]) and len(outs[0]) >= 1
    for a, b in zip(*outs):
        assert bool(torch.isfinite(b).all()), "a kernel read outside a buffer"
        assert torch.equal(a, b)
