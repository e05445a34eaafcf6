"""The opt-in precision mode "bf16x3" (csrc/tcn_split.hip; blocks.set_precision): the temporal conv on the bf16 matrix
pipe with fp32 operands split into three bf16 pieces.  It is fp32-GRADE, not exact fp32, so it has tests of its own: every
golden fixture that contains a 9 x 1 temporal conv (G3-G6, G8) and the full-size configs, against the reference's outputs /
the oracle at the SAME 1e-4 absolute tolerance, with the achieved error on record (tests/helpers.check_parity ->
parity report, mode = "bf16x3").  The default mode must not change by a bit when the mode is switched back."""
import pytest
import torch

import _bootstrap
from oracle import stgcn_oracle as o
from tests.helpers import BLOCK_OUT_KEYS, check_parity, g6_state_dict, g8_state_dict, load_golden, unit_scale_

pytestmark = pytest.mark.gpu
pkg = _bootstrap.load()
DEV = "cuda:0"
MODE = "bf16x3"


def _A(v=25):
    return (pkg.ntu_graph() if v == 25 else pkg.kinetics_graph()).A


@pytest.mark.parametrize("tag", ["nores", "ident", "convres", "strided", "nopad", "nopad_strided"])
def test_block_golden_bf16x3(tag):
    a, sd = load_golden(f"g3_block_{tag}")
    ci, co, s, res, tp = (int(v) for v in a["meta"])
    m = pkg.SpatioTemporalBlock(ci, co, _A(), s, bool(res), temporal_padding=tp).eval()
    m.load_state_dict(sd, strict=True)
    pkg.set_precision(m, MODE)
    y = m.to(DEV)(torch.from_numpy(a["x"]).to(DEV))
    check_parity(y.cpu(), a["y"], mode=MODE)


def test_stack_and_config1_golden_bf16x3():
    from closed_form import closed_form_input

    a, sd = load_golden("g4_stack")
    A = _A()
    stack = torch.nn.Sequential(pkg.SpatioTemporalBlock(3, 3, A, residual=False), pkg.SpatioTemporalBlock(3, 3, A),
                                pkg.SpatioTemporalBlock(3, 4, A, stride=2)).eval()
    stack.load_state_dict(sd, strict=True)
    pkg.set_precision(stack, MODE)
    check_parity(stack.to(DEV)(torch.from_numpy(a["x"]).to(DEV)).cpu(), a["y"], mode=MODE)
    a, sd = load_golden("g5_config1_block")                       # BASELINE config 1
    m = pkg.SpatioTemporalBlock(3, 64, A, residual=False).eval()
    m.load_state_dict(sd, strict=True)
    pkg.set_precision(m, MODE)
    y = m.to(DEV)(torch.from_numpy(closed_form_input((2, 3, 300, 25), salt=5.0)).to(DEV)).cpu()
    check_parity(y.reshape(-1)[::7], a["y_sub7"], mode=MODE)


@pytest.mark.parametrize("tag", ["ntu", "kin"])
def test_full_stgcn_golden_bf16x3(tag):
    """The reference's whole StGcn (fixture G6): logits and layer taps after 1 / 5 / 8 / 10 blocks of split arithmetic."""
    a, sd, x = g6_state_dict(tag)
    v, classes = (25, 60) if tag == "ntu" else (18, 400)
    net = pkg.StGcn(_A(v), input_shape=(3, 300, v, 2), num_classes=classes).eval()
    net.load_state_dict(sd, strict=True)
    pkg.set_precision(net, MODE)
    net = net.to(DEV)
    taps = {}
    hooks = [net.layers[f"layer{i}"].register_forward_hook(lambda m, inp, out, i=i: taps.__setitem__(i, out))
             for i in (1, 5, 8, 10)]
    logits = net(x.to(DEV)).cpu()
    for h in hooks:
        h.remove()
    for i in (1, 5, 8, 10):
        check_parity(taps[i].cpu().reshape(-1)[::997], a[f"layer{i}_sub"], mode=MODE, note=f"layer{i}")
    check_parity(logits, a["logits"], mode=MODE)


def test_full_agcn_golden_bf16x3():
    a, sd, x = g8_state_dict()
    net = pkg.AGcn(_A(18), (3, 300, 18, 2), 400).eval()
    net.load_state_dict(sd, strict=True)
    pkg.set_precision(net, MODE)
    check_parity(net.to(DEV)(x.to(DEV)).cpu(), a["logits"], mode=MODE)


@pytest.mark.parametrize("ci,co,stride,res,T,N,v", [
    (64, 64, 1, True, 37, 3, 25),      # identity residual, ragged last tile, 64-row tiles
    (64, 128, 2, True, 50, 2, 25),     # strided conv residual: 2-tap weight stages
    (128, 256, 2, True, 31, 2, 25),    # odd T with stride 2, two M tiles
    (256, 256, 1, True, 20, 2, 25),
    (3, 64, 1, False, 300, 1, 25),     # layer-1 shape
    (20, 12, 1, True, 9, 1, 25),       # channel counts off every tile / chunk multiple
    (130, 70, 2, True, 13, 2, 18),
    (64, 64, 1, True, 11, 7, 18),
    (4, 4, 1, True, 1, 3, 25),         # a single frame
])
def test_block_vs_oracle_seeded_bf16x3(ci, co, stride, res, T, N, v):
    g = torch.Generator().manual_seed(4321 + ci + co + T)
    m = pkg.SpatioTemporalBlock(ci, co, _A(v), stride, res).eval()
    with torch.no_grad():
        for name, prm in m.named_parameters():
            if name.endswith("graph_attn") or name.endswith("bn.weight") or name.endswith("residual.1.weight"):
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
    m = m.to(DEV)
    exact = m(x.to(DEV)).cpu()
    pkg.set_precision(m, MODE)
    got = m(x.to(DEV)).cpu()
    check_parity(got, want, mode=MODE, shape=(ci, co, stride, res, T, N, v))
    check_parity(got, exact, mode=MODE, note="vs the exact-fp32 kernels")
    pkg.set_precision(m, "f32")
    assert torch.equal(m(x.to(DEV)).cpu(), exact)                    # the default mode is untouched by the round trip


def test_config2_batch256_clip_forward_bf16x3():
    """BASELINE configs[1] at full size in the split mode: logits of an 8-clip slice vs the oracle, batch invariance."""
    a, sd, _ = g6_state_dict("ntu")
    net = pkg.StGcn(_A()).eval()
    net.load_state_dict(sd, strict=True)
    pkg.set_precision(net, MODE)
    net = net.to(DEV)
    x = torch.rand((256, 3, 300, 25, 2), device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    full = net(x)
    assert full.shape == (256, 60) and bool(torch.isfinite(full).all())
    idx = [0, 1, 37, 100, 128, 200, 254, 255]
    with torch.no_grad():
        want = o.stgcn_forward(x[idx].cpu(), sd)
    check_parity(full[idx].cpu(), want, mode=MODE)
    part = net(x[64:128].contiguous())
    assert torch.equal(part, full[64:128])


@pytest.mark.parametrize("ci,co,T,N,v", [(64, 128, 33, 2, 25), (128, 128, 40, 3, 25), (128, 256, 21, 2, 25), (256, 256, 17, 2, 18),
                                         (70, 128, 9, 1, 25), (256, 256, 1, 5, 25)])
def test_graph_conv_vs_oracle_bf16x3(ci, co, T, N, v):
    """csk_gcn_stage_bf16x3 (channel mix of the graph conv in split arithmetic, 128-row tiles): identity and conv
    gcn_residual, ragged tiles, odd channel counts, vs the oracle and vs the exact kernel."""
    from tests.helpers import GCN_OUT_KEYS
    g = torch.Generator().manual_seed(77 + ci + co + T)
    m = pkg.GraphConvolution(ci, co, _A(v)).eval()
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
    sd = {k: t.clone() for k, t in m.state_dict().items()}
    x = torch.rand(N, ci, T, v, generator=g)
    want = unit_scale_(m, sd, lambda s: o.graph_conv(x, s), GCN_OUT_KEYS)
    m = m.to(DEV)
    exact = m(x.to(DEV)).cpu()
    m.precision = MODE
    m.refold()
    assert m._split_applies(m._packed_ops(torch.device(DEV)))
    got = m(x.to(DEV)).cpu()
    check_parity(got, want, mode=MODE, shape=(ci, co, T, N, v))
    check_parity(got, exact, mode=MODE, note="vs the exact-fp32 kernel")
    m.precision = "f32"
    m.refold()
    assert torch.equal(m(x.to(DEV)).cpu(), exact)


def test_continual_stepping_refuses_bf16x3_but_clip_forward_works():
    """The split kernels exist for the clip form only: a continual block in bf16x3 mode computes ``forward`` (clip) with
    them and refuses to step (no silent fall-back to another arithmetic)."""
    a, sd = load_golden("g3_block_ident")
    ci, co, s, res, tp = (int(v) for v in a["meta"])
    blk = pkg.CoSpatioTemporalBlock(ci, co, _A(), stride=s, residual=bool(res), padding="equal").eval()
    blk.load_state_dict(sd, strict=True)
    pkg.set_precision(blk, MODE)
    blk = blk.to(DEV)
    x = torch.from_numpy(a["x"]).to(DEV)
    check_parity(blk.forward(x).cpu(), a["y"], mode=MODE)
    with pytest.raises(NotImplementedError, match="clip kernels only"):
        blk.forward_step(x[:, :, 0].contiguous())
    pkg.set_precision(blk, "f32")
    assert blk.forward_step(x[:, :, 0].contiguous()) is None       # back to the default: stepping works (no output yet)


def test_costgcn_model_refuses_to_step_in_bf16x3_with_either_engine():
    a, sd, x = g6_state_dict("ntu")
    for native_plan in (True, False):
        co = pkg.CoStGcn(_A(), pool_size=4, pool_padding=1).eval()
        co.use_native_plan = native_plan
        co.load_state_dict(sd, strict=True)
        co = co.to(DEV)
        f = x[:1, :, 0].contiguous().to(DEV)
        assert co.forward_step(f) is None                        # default precision: steps
        pkg.set_precision(co, MODE)
        with pytest.raises(NotImplementedError, match="clip kernels only"):
            co.forward_step(f)
        check_parity(co.forward(x[:1].to(DEV)).cpu(), pkg.set_precision(co, "f32").forward(x[:1].to(DEV)).cpu(), mode=MODE,
                     note="CoStGcn.forward (clip form) in bf16x3 vs f32")


def test_split_kernels_are_deterministic_across_launches():
    """Race screen of the new pipelines (ping-pong weight stages, single-buffered activation tile, aggregate-then-MFMA of
    the graph-conv split kernel, partial-logit attention): the same inputs must give the same bits on every launch, at a
    size where every CU runs several tiles back to back."""
    a, sd, x = g6_state_dict("ntu")
    net = pkg.StGcn(_A()).eval()
    net.load_state_dict(sd, strict=True)
    pkg.set_precision(net, MODE)
    net = net.to(DEV)
    xb = torch.rand((48, 3, 300, 25, 2), device=DEV, generator=torch.Generator(device=DEV).manual_seed(9))
    ref = net(xb)
    for _ in range(12):
        assert torch.equal(net(xb), ref)
    ag = pkg.AGcn(_A(18), (3, 300, 18, 2), 400).eval()
    a8, sd8, x8 = g8_state_dict()
    ag.load_state_dict(sd8, strict=True)
    ag = ag.to(DEV)
    xk = torch.rand((24, 3, 300, 18, 2), device=DEV, generator=torch.Generator(device=DEV).manual_seed(10))
    ref = ag(xk)
    for _ in range(8):
        assert torch.equal(ag(xk), ref)


def test_precision_argument_errors():
    m = pkg.SpatioTemporalBlock(4, 4, _A()).eval()
    with pytest.raises(ValueError, match="precision must be"):
        pkg.set_precision(m, "bf16")
    with pytest.raises(ValueError, match="no SpatioTemporalBlock"):
        pkg.set_precision(pkg.GraphConvolution(4, 4, _A()), MODE)
    lib = pkg.native.lib()
    rc = lib.csk_tcn_stage_bf16x3(None, None, None, None, None, None, 1, 1, 1, 1, 25, 9, 1, 4, 0, 0, 0, 0, 1, None)
    assert rc == -1 and b"null pointer" in lib.csk_last_error()
