"""Edge cases of the clip and continual block paths vs the oracle: very short and very long clips, odd lengths
under stride 2, single channels, Kinetics joint count, odd skeleton counts, and argument errors."""
import pytest
import torch

import _bootstrap
from oracle import stgcn_oracle as o
from tests.helpers import max_err

pytestmark = pytest.mark.gpu
pkg = _bootstrap.load()
DEV = "cuda:0"
TOL = 1e-4


def _rand_block(ci, co, stride, res, v, seed, padding=-1):
    A = (pkg.ntu_graph() if v == 25 else pkg.kinetics_graph()).A
    m = pkg.SpatioTemporalBlock(ci, co, A, stride, res, temporal_padding=padding).eval()
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, prm in m.named_parameters():
            if name.endswith("graph_attn") or ("bn" in name and name.endswith("weight")) or name.endswith("residual.1.weight"):
                prm.copy_(torch.rand(prm.shape, generator=g) + 0.5)
            elif name.endswith("bias"):
                prm.copy_(torch.rand(prm.shape, generator=g) - 0.5)
        for name, buf in m.named_buffers():
            if name.endswith("running_var"):
                buf.copy_(torch.rand(buf.shape, generator=g) + 0.5)
            elif name.endswith("running_mean"):
                buf.copy_(torch.rand(buf.shape, generator=g) - 0.5)
    return m, {k: t.clone() for k, t in m.state_dict().items()}


@pytest.mark.parametrize("ci,co,stride,res,T,N,v", [
    (4, 4, 1, True, 1, 3, 25),        # a single frame
    (4, 8, 2, True, 1, 1, 25),        # single frame, stride 2, conv residual
    (8, 8, 1, True, 2, 5, 25),
    (8, 16, 2, True, 7, 3, 18),       # odd T under stride 2, Kinetics joints, odd skeleton count
    (1, 1, 1, True, 11, 2, 25),       # single channel
    (3, 64, 1, False, 1000, 1, 18),   # long clip: many ragged tiles
    (64, 64, 1, True, 11, 7, 18),
    (130, 70, 2, True, 13, 2, 25),    # channel counts off every tile / chunk multiple
])
def test_block_edge_shapes(ci, co, stride, res, T, N, v):
    m, sd = _rand_block(ci, co, stride, res, v, seed=ci * 7 + T)
    x = torch.rand(N, ci, T, v, generator=torch.Generator().manual_seed(T))
    with torch.no_grad():
        want = o.st_block(x, sd, "", stride, res)
    got = m.to(DEV)(x.to(DEV)).cpu()
    assert got.shape == want.shape
    assert max_err(got, want) <= TOL * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("T,pad", [(9, 0), (10, 0), (12, 2)])
def test_unpadded_block_minimal_lengths(T, pad):
    """temporal_padding < 4: the clip must be at least k - 2p frames long; output length T + 2p - 8."""
    m, sd = _rand_block(4, 4, 1, True, 25, seed=T, padding=pad)
    x = torch.rand(2, 4, T, 25, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        want = o.st_block(x, sd, "", 1, True, pad)
    got = m.to(DEV)(x.to(DEV)).cpu()
    assert got.shape == want.shape == (2, 4, T + 2 * pad - 8, 25) and max_err(got, want) <= TOL


def test_too_short_clip_is_an_error():
    m, _ = _rand_block(4, 4, 1, True, 25, seed=0, padding=0)
    with pytest.raises(RuntimeError, match="shorter than kernel"):
        m.to(DEV)(torch.rand(1, 4, 8, 25, device=DEV))
    with pytest.raises(RuntimeError):
        m.to(DEV)(torch.rand(1, 5, 20, 25, device=DEV))            # wrong channel count
    with pytest.raises(RuntimeError):
        m.to(DEV)(torch.rand(1, 4, 20, 18, device=DEV))            # wrong joint count for the adjacency


@pytest.mark.parametrize("n_streams,v", [(1, 25), (3, 18), (7, 25)])
def test_continual_block_odd_stream_counts(n_streams, v):
    """P = streams * V is padded to a multiple of 4 inside the state slab; results must not depend on it."""
    A = (pkg.ntu_graph() if v == 25 else pkg.kinetics_graph()).A
    ref, sd = _rand_block(6, 6, 1, True, v, seed=v + n_streams)
    co = pkg.CoSpatioTemporalBlock(6, 6, A, padding=4).eval()
    co.load_state_dict(sd, strict=True)
    co = co.to(DEV)
    x = torch.rand(n_streams, 6, 23, v, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        want = o.st_block(x, sd, "", 1, True)
    got = co.forward_steps(x.to(DEV), pad_end=True).cpu()
    assert got.shape == want.shape and max_err(got, want) <= TOL


def test_stream_shards_get_concurrent_hardware_queues():
    """HIP maps streams onto a few hardware queues; shards on one queue would serialise.  The probe must see a
    stream as serial with itself, and the shard streams picked must overlap each other and the current stream."""
    from continual_skeletons_amd import parallel
    cur = torch.cuda.current_stream(DEV)
    s = torch.cuda.Stream(device=DEV)
    assert not parallel.streams_overlap(s, s)
    picked = parallel.concurrent_streams(2, DEV)
    assert len(picked) == 2 and parallel.streams_overlap(picked[0], picked[1])
    assert all(parallel.streams_overlap(p, cur) for p in picked)
    with pytest.raises(RuntimeError, match="spin_us"):
        parallel.streams_overlap(s, cur, spin_us=1)
