"""Edge cases of the clip and continual block paths vs the oracle: very short and very long clips, odd lengths
under stride 2, single channels, Kinetics joint count, odd skeleton counts, and argument errors."""
import pytest
import torch

import _bootstrap
from oracle import stgcn_oracle as o
from tests.helpers import BLOCK_OUT_KEYS, check_parity, max_err, unit_scale_

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
    want = unit_scale_(m, sd, lambda s: o.st_block(x, s, "", stride, res), BLOCK_OUT_KEYS)
    got = m.to(DEV)(x.to(DEV)).cpu()
    check_parity(got, want, shape=(ci, co, stride, res, T, N, v))


@pytest.mark.parametrize("c,co,stride,v,T", [
    (128, 128, 2, 40, 21),     # 128-row tiles narrowed (span of a stride-2 tile exceeds 9 sweeps of 64 positions)
    (128, 128, 2, 64, 13),     # V = 64: the 64-row kernel (14 sweeps) keeps half its columns, the 128-row one would keep 1
    (16, 128, 3, 48, 17),      # stride 3
    (8, 64, 2, 64, 9),         # 64-row tiles narrowed
])
def test_temporal_conv_wide_skeletons_narrowed_tiles(c, co, stride, v, T):
    """csk_tcn_stage_f32 on shapes whose activation span per tile exceeds the register staging (V > 32 with stride >= 2):
    the tile is narrowed / the other tile height is chosen (csrc/tcn.hip); results must not depend on that choice."""
    torch.manual_seed(c + v)
    m = pkg.TemporalConvolution(c, co, 9, stride, 4).eval()
    with torch.no_grad():
        m.t_conv.weight.mul_(0.3)
        m.bn.weight.uniform_(0.5, 1.5); m.bn.bias.uniform_(-0.5, 0.5)
        m.bn.running_mean.uniform_(-0.5, 0.5); m.bn.running_var.uniform_(0.5, 1.5)
    sd = {k: t.clone() for k, t in m.state_dict().items()}
    x = torch.rand(2, c, T, v)
    with torch.no_grad():
        want = o.temporal_conv(x, sd, "", stride, 4)
    got = m.to(DEV)(x.to(DEV)).cpu()
    check_parity(got, want, shape=(c, co, stride, v, T))


@pytest.mark.parametrize("T,pad", [(9, 0), (10, 0), (12, 2)])
def test_unpadded_block_minimal_lengths(T, pad):
    """temporal_padding < 4: the clip must be at least k - 2p frames long; output length T + 2p - 8."""
    m, sd = _rand_block(4, 4, 1, True, 25, seed=T, padding=pad)
    x = torch.rand(2, 4, T, 25, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        want = o.st_block(x, sd, "", 1, True, pad)
    got = m.to(DEV)(x.to(DEV)).cpu()
    assert got.shape == want.shape == (2, 4, T + 2 * pad - 8, 25)
    check_parity(got, want)


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
    assert got.shape == want.shape
    check_parity(got, want)


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


def _random_adjacency(v, dense, g):
    """(3, V, V): dense random weights (general GCN kernel) or a skeleton-like pattern with <= 1 / 1 / 4 non-zeros
    per column (sparse kernel), random positions and values."""
    if dense:
        return torch.rand(3, v, v, generator=g) / v
    A = torch.zeros(3, v, v)
    for w in range(v):
        A[0, w, w] = float(torch.rand(1, generator=g)) + 0.5
        A[1, int(torch.randint(v, (1,), generator=g)), w] = float(torch.rand(1, generator=g)) + 0.1
        for src in torch.randperm(v, generator=g)[: int(torch.randint(1, min(4, v) + 1, (1,), generator=g))]:
            A[2, int(src), w] = float(torch.rand(1, generator=g)) + 0.1
    return A


@pytest.mark.parametrize("seed", range(16))
def test_block_random_sweep(seed):
    """Seeded random blocks: channel counts off every padding granule, strides 1-3, clip lengths 1-90, 2-40 joints,
    random adjacencies of both kinds, all three residual forms -- HIP block vs the oracle."""
    g = torch.Generator().manual_seed(1000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    v, stride, res = ri(2, 40), ri(1, 3), bool(ri(0, 3))
    ci = ri(1, 150)
    co = ci if (res and ri(0, 1)) else ri(1, 150)
    T, N = ri(1, 90), ri(1, 4)
    A = _random_adjacency(v, dense=bool(seed % 2), g=g)
    m = pkg.SpatioTemporalBlock(ci, co, A, stride, res).eval()
    with torch.no_grad():
        for name, prm in m.named_parameters():
            if name.endswith("graph_attn") or ("bn" in name and name.endswith("weight")):
                prm.copy_(torch.rand(prm.shape, generator=g) + 0.5)
            elif name.endswith("bias"):
                prm.copy_(torch.rand(prm.shape, generator=g) - 0.5)
        for name, buf in m.named_buffers():
            if name.endswith("running_var"):
                buf.copy_(torch.rand(buf.shape, generator=g) + 0.5)
            elif name.endswith("running_mean"):
                buf.copy_(torch.rand(buf.shape, generator=g) - 0.5)
    sd = {k: t.clone() for k, t in m.state_dict().items()}
    x = torch.rand(N, ci, T, v, generator=g)
    want = unit_scale_(m, sd, lambda s: o.st_block(x, s, "", stride, res), BLOCK_OUT_KEYS)
    got = m.to(DEV)(x.to(DEV)).cpu()
    check_parity(got, want, shape=(ci, co, stride, res, T, N, v))


@pytest.mark.parametrize("seed", range(8))
def test_continual_block_random_sweep(seed):
    """Seeded random continual blocks: stepping the whole clip with pad_end reproduces the clip oracle
    (tests/test_cost_gcn.py:66-68) for random channel counts, stride 1-2, stream counts, joint counts and
    adjacencies of both kinds."""
    g = torch.Generator().manual_seed(2000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    v, stride, res = ri(2, 33), ri(1, 2), bool(ri(0, 3))
    ci = ri(1, 140)
    co = ci if (res and ri(0, 1)) else ri(1, 140)
    T, N = ri(9, 40), ri(1, 5)
    A = _random_adjacency(v, dense=bool(seed % 2), g=g)
    ref = pkg.SpatioTemporalBlock(ci, co, A, stride, res).eval()
    with torch.no_grad():
        for name, prm in ref.named_parameters():
            if name.endswith("graph_attn") or ("bn" in name and name.endswith("weight")):
                prm.copy_(torch.rand(prm.shape, generator=g) + 0.5)
            elif name.endswith("bias"):
                prm.copy_(torch.rand(prm.shape, generator=g) - 0.5)
        for name, buf in ref.named_buffers():
            if name.endswith("running_var"):
                buf.copy_(torch.rand(buf.shape, generator=g) + 0.5)
            elif name.endswith("running_mean"):
                buf.copy_(torch.rand(buf.shape, generator=g) - 0.5)
    sd = {k: t.clone() for k, t in ref.state_dict().items()}
    x = torch.rand(N, ci, T, v, generator=g)
    want = unit_scale_(ref, sd, lambda s: o.st_block(x, s, "", stride, res), BLOCK_OUT_KEYS)
    blk = pkg.CoSpatioTemporalBlock(ci, co, A, stride=stride, residual=res, padding=4).eval()
    blk.load_state_dict(sd, strict=True)
    got = blk.to(DEV).forward_steps(x.to(DEV), pad_end=True).cpu()
    check_parity(got, want, shape=(ci, co, stride, res, T, N, v))


def test_nan_propagates_through_relu_epilogues():
    """torch.relu / np.maximum in the reference propagate NaN; the fused ReLU epilogues must not turn a NaN activation
    into 0 (fmaxf would).  One NaN input element must surface as NaN in the outputs that depend on it, in clip and in
    step mode, and everything else stays finite."""
    A = pkg.ntu_graph().A
    blk = pkg.SpatioTemporalBlock(4, 4, A).eval().to(DEV)
    x = torch.rand(1, 4, 12, 25, device=DEV)
    x[0, 2, 5, 7] = float("nan")
    y = blk(x)
    assert bool(torch.isnan(y[0, :, 5, 7]).all())                       # identity residual carries it straight through
    assert bool(torch.isnan(y).any()) and bool(torch.isfinite(y[0, :, 0, 0]).all())
    co = pkg.CoSpatioTemporalBlock(4, 4, A, padding=4).eval().to(DEV)
    outs = []
    for t in range(12):
        o_ = co.forward_step(x[:, :, t].contiguous())
        if o_ is not None:
            outs.append(o_)
    z = torch.stack(outs, dim=2)                                         # step s emits clip frame s - 4
    assert bool(torch.isnan(z[0, :, 5, 7]).all()) and bool(torch.isfinite(z[0, :, 0, 0]).all())


@pytest.mark.parametrize("c,res", [(4, 1), (8, 1), (8, 0), (12, 1), (6, 1), (16, 1)])
def test_tcn_step_never_reads_behind_the_last_ring_slot(c, res):
    """Channel counts that are whole 4-channel chunks but not whole CSK_CPAD = 16 blocks: the packed weights have padding rows,
    the state ring has none.  The ring sits at the END of a buffer whose tail is NaN (then zero): a kernel that walks the padding
    rows reads them behind the last slot -- NaN x 0 = NaN in the output; on a box where the ring ends at the end of a mapping the
    same read is a GPU memory fault (seen in round 6: csk_tcn_step_f32's fast instantiation took C = 4 and C = 8)."""
    from continual_skeletons_amd import native
    lib = native.lib()
    P, slots = 176, 12
    tc = pkg.TemporalConvolution(c, c, kernel_size=9, stride=1, padding=4).eval()
    import bench
    bench.randomise_(tc, 3)
    tc = tc.to(DEV)
    ops = tc._packed_ops(torch.device(DEV))
    g = torch.Generator().manual_seed(c)
    ring_h = torch.rand((slots, c, P), generator=g)
    xres_h = torch.rand((8, c, P), generator=g)
    outs = []
    for fill in (float("nan"), 0.0):
        buf = torch.full((slots * c * P + 16 * P,), fill, device=DEV)
        ring = buf[: slots * c * P].view(slots, c, P)
        ring.copy_(ring_h)
        xbuf = torch.full((8 * c * P + 16 * P,), fill, device=DEV)
        xres = xbuf[: 8 * c * P].view(8, c, P)
        xres.copy_(xres_h)
        out = torch.zeros((4, c, P), device=DEV)
        # head = slots - 1: the newest frame is in the LAST physical slot (the window is slots 3 .. 11); xres slot 7 likewise
        rc = lib.csk_tcn_step_f32(native.ptr(ring), slots, slots - 1, 1, 1, native.ptr(ops["w"]), native.ptr(xres) if res else None, 8, 7, 1,
                                  None, native.ptr(ops["bias"]), native.ptr(out), 4, 0, c, c, P, 9, res, c if res else 0, 1, 1, None,
                                  native.stream_of(ring))
        native.check(rc, "csk_tcn_step_f32")
        torch.cuda.synchronize()
        outs.append(out[0].cpu())
    assert bool(torch.isfinite(outs[0]).all()), "the kernel read rows behind the last ring slot"
    assert torch.equal(outs[0], outs[1])
    # and the value: the module's clip forward over the 9 window frames (slots 3 .. 11), centre frame, + identity residual, ReLU
    x = ring_h[3:12].view(9, c, 8, 22).permute(2, 1, 0, 3).contiguous()          # positions as 8 "samples" x 22 "joints": (N, C, T, V)
    clip = tc(x.to(DEV))[:, :, 4].permute(1, 0, 2).reshape(c, P).cpu()
    want = torch.relu(clip + (xres_h[7] if res else 0))
    check_parity(outs[0], want)


def _guarded(t, fill, pad=1 << 16):
    """A copy of `t` inside a larger buffer whose `pad` elements in FRONT of it and behind it hold `fill` (NaN: a kernel that reads
    outside its operand -- against a zero weight, or for a lane it later masks -- shows up as NaN in the output; on a box where the
    operand ends a mapping the same read is a memory fault)."""
    buf = torch.full((t.numel() + 2 * pad,), fill, device=DEV)
    v = buf[pad: pad + t.numel()].view(t.shape)
    v.copy_(t)
    return v


@pytest.mark.parametrize("ci,co,frames,skel", [(64, 64, 4, 2048), (3, 64, 4, 2048), (128, 256, 2, 2048), (12, 24, 4, 2048),
                                               (64, 64, 4, 7), (6, 6, 1, 3), (256, 256, 1, 2047)])
def test_graph_conv_step_shapes_read_only_their_input(ci, co, frames, skel):
    """csk_gcn_stage_f32 on channel-major frames (the continual engine's call): at 2048 skeletons the slot-balanced tiles
    (gcn16_kernel), at small sizes the 32x32x2 kernel; ragged channel counts and a ragged last tile.  The input sits between NaN
    guards: the output must be finite and the same as with zero guards."""
    torch.manual_seed(ci + co)
    g = pkg.GraphConvolution(ci, co, pkg.ntu_graph().A).eval()
    import bench
    bench.randomise_(g, 1)
    g = g.to(DEV)
    P = (skel * 25 + 3) // 4 * 4
    x_h = torch.rand((frames, ci, P))
    outs = []
    for fill in (float("nan"), 0.0):
        x = _guarded(x_h.to(DEV), fill)
        y = torch.zeros((frames, co, P), device=DEV)
        g.stage(x, y, n_seg=frames, frames=skel, x_strides=(ci * P, P), y_strides=(co * P, P))
        torch.cuda.synchronize()
        outs.append(y[:, :, : skel * 25].cpu())
    assert bool(torch.isfinite(outs[0]).all()), "the kernel read outside x"
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("c,co,t,stride,v", [(64, 64, 20, 1, 25), (64, 128, 21, 2, 25), (64, 64, 16, 1, 25), (128, 128, 37, 1, 18),
                                             (256, 256, 9, 1, 18), (128, 256, 30, 2, 18), (8, 8, 20, 1, 25), (128, 128, 20, 1, 25)])
def test_temporal_conv_clip_shapes_read_only_their_input(c, co, t, stride, v):
    """csk_tcn_stage_f32 (16-wide tile family where the policy takes it, the 32x32x2 kernel elsewhere): frame counts that are
    not whole tiles, stride 2 with an odd count, the zero padding at both ends.  Input between NaN guards: finite output, equal
    to the run with zero guards."""
    torch.manual_seed(c + t)
    tc = pkg.TemporalConvolution(c, co, kernel_size=9, stride=stride, padding=4).eval()
    import bench
    bench.randomise_(tc, 2)
    tc = tc.to(DEV)
    x_h = torch.rand((3, c, t, v))
    outs = []
    for fill in (float("nan"), 0.0):
        outs.append(tc(_guarded(x_h.to(DEV), fill)).cpu())
    assert bool(torch.isfinite(outs[0]).all()), "the kernel read outside its input"
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("ci,co,mode", [(3, 64, "step"), (64, 64, "step"), (128, 256, "step"), (64, 128, "clip"), (3, 64, "clip")])
def test_adaptive_graph_conv_reads_only_its_input(ci, co, mode):
    """A-GCN / CoAGCN (Kinetics joints): embedding + attention + dense graph conv launches with the input between NaN guards --
    step form on channel-major frames (511 skeletons: a ragged last tile), clip form on (N, C, T, V) with T = 23."""
    torch.manual_seed(ci)
    A = pkg.kinetics_graph().A
    g = pkg.AdaptiveGraphConvolution(ci, co, A).eval()
    import bench
    bench.randomise_(g, 4, attn_scale=1 / 18)
    g = g.to(DEV)
    outs = []
    if mode == "step":
        skel, frames = 511, 2
        P = skel * 18 + 2
        x_h = torch.rand((frames, ci, P))
        for fill in (float("nan"), 0.0):
            x = _guarded(x_h.to(DEV), fill)
            y = torch.zeros((frames, co, P), device=DEV)
            g.stage(x, y, n_seg=frames, frames=skel, x_strides=(ci * P, P), y_strides=(co * P, P))
            torch.cuda.synchronize()
            outs.append(y[:, :, : skel * 18].cpu())
    else:
        x_h = torch.rand((3, ci, 23, 18))
        for fill in (float("nan"), 0.0):
            outs.append(g(_guarded(x_h.to(DEV), fill)).cpu())
    assert bool(torch.isfinite(outs[0]).all()), "a kernel read outside x"
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("ci,co,res,stride", [(4, 4, True, 1), (8, 8, True, 1), (12, 12, False, 1), (3, 8, False, 1), (6, 6, True, 1),
                                              (64, 64, True, 1), (8, 16, True, 2), (64, 128, True, 2)])
def test_block_step_with_the_rings_between_nan_guards(ci, co, res, stride, fuse):
    """A continual block's cycle (graph conv of the new frames + emitting temporal steps; fused and as two launches) with its
    three rings re-seated between NaN guards: the rings and the emissions must be the ones of the block whose rings sit between
    zero guards -- no launch reads a row outside a ring (channel counts with weight padding rows: 4, 8, 12; 7 skeletons in 176
    positions: a ragged tile; stride 2: emissions every other frame)."""
    import bench
    blocks, states = [], []
    for fill in (float("nan"), 0.0):
        torch.manual_seed(1)
        b = pkg.CoSpatioTemporalBlock(ci, co, pkg.ntu_graph().A, stride, residual=res, padding="equal").eval()
        bench.randomise_(b, 2)
        b = b.to(DEV)
        b.fuse_step = fuse
        st = b.bind_state(176, torch.device(DEV))
        st.y, st.out, st.xin = _guarded(st.y, fill), _guarded(st.out, fill), _guarded(st.xin, fill)
        blocks.append(b)
        states.append(st)
    g = torch.Generator().manual_seed(4)
    got = None
    for cyc in range(9):
        x = torch.rand((4, ci, 176), generator=g).to(DEV)
        rets = []
        for b, st in zip(blocks, states):
            for f in range(4):
                st.xin[(st.s + f) % st.xin.shape[0]] = x[f]
            rets.append(b.engine_advance(4, 7, 25))
        torch.cuda.synchronize()
        assert rets[0] == rets[1]
        for name in ("y", "out"):
            a_, b_ = getattr(states[0], name), getattr(states[1], name)
            assert bool(torch.isfinite(a_).all()), (cyc, name, "a launch read outside a ring")
            assert torch.equal(a_, b_), (cyc, name)
        got = rets[0]
    assert got is not None
