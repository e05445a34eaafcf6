"""GPU parity of the A-GCN adaptive graph convolution (BASELINE.json configs[3]): golden vectors from the
reference's AdaptiveGraphConvolution (fixture G7), seeded comparisons against the oracle at real channel counts,
and the CoAGCN per-frame (T = 1) use inside a continual block."""
import pytest
import torch

import _bootstrap
from oracle import stgcn_oracle as o
from tests.helpers import GCN_OUT_KEYS, check_parity, g8_state_dict, load_golden, randomise_unit_, unit_scale_

pytestmark = pytest.mark.gpu
TOL = 1e-4
pkg = _bootstrap.load()
DEV = "cuda:0"
A_KIN = pkg.kinetics_graph().A


@pytest.mark.parametrize("tag,ci,co", [("neq", 3, 8), ("eq", 8, 8)])
@pytest.mark.parametrize("t", [1, 6])
def test_adaptive_graph_conv_golden(tag, ci, co, t):
    a, sd = load_golden(f"g7_agcn_{tag}")
    m = pkg.AdaptiveGraphConvolution(ci, co, A_KIN).eval()
    m.load_state_dict(sd, strict=True)
    y = m.to(DEV)(torch.from_numpy(a[f"x_t{t}"]).to(DEV))
    check_parity(y.cpu(), a[f"y_t{t}"])


def _randomise(m, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, prm in m.named_parameters():
            if name.endswith("graph_attn") or ("bn" in name and name.endswith("weight")) or name.endswith("residual.1.weight"):
                prm.copy_(torch.rand(prm.shape, generator=g) + 0.5)
            elif name.endswith("bias"):
                prm.copy_(torch.rand(prm.shape, generator=g) - 0.5)
            elif "a_conv" in name or "b_conv" in name:
                prm.copy_(torch.randn(prm.shape, generator=g) * 0.5)       # make the attention non-uniform
        for name, buf in m.named_buffers():
            if name.endswith("running_var"):
                buf.copy_(torch.rand(buf.shape, generator=g) + 0.5)
            elif name.endswith("running_mean"):
                buf.copy_(torch.rand(buf.shape, generator=g) - 0.5)


# V = 18: gcn_stage_dense2_kernel (on-the-fly dense aggregation; 64- and 128-row tiles, identity and conv residual, a last tile
# of 1 ... 6 frames, C_out below the tile height), also at V = 25 (odd: joints staged one by one, half-empty last pair)
@pytest.mark.parametrize("ci,co,t,v", [(64, 64, 40, 18), (64, 128, 17, 18), (3, 64, 300, 18), (16, 16, 9, 25), (128, 128, 10, 18),
                                       (256, 256, 15, 18), (16, 24, 9, 18), (128, 256, 1, 18), (40, 40, 7, 18),
                                       (128, 128, 11, 25), (64, 128, 7, 25), (256, 256, 4, 25)])
def test_adaptive_graph_conv_vs_oracle(ci, co, t, v):
    A = A_KIN if v == 18 else pkg.ntu_graph().A
    m = pkg.AdaptiveGraphConvolution(ci, co, A).eval()
    _randomise(m, 7 + ci + co)
    sd = {k: v_.clone() for k, v_ in m.state_dict().items()}
    x = torch.rand(3, ci, t, v, generator=torch.Generator().manual_seed(5))
    want = unit_scale_(m, sd, lambda s: o.adaptive_graph_conv(x, s), GCN_OUT_KEYS)
    got = m.to(DEV)(x.to(DEV)).cpu()
    check_parity(got, want, shape=(ci, co, t, v))


@pytest.mark.parametrize("ci,co,t,v,scale", [(64, 64, 12, 18, 6.0), (64, 64, 1, 18, 6.0), (128, 128, 1, 18, 4.0), (64, 128, 9, 25, 6.0),
                                             (64, 64, 1, 25, 8.0)])
def test_adaptive_graph_conv_peaked_attention(ci, co, t, v, scale):
    """The softmax of every attention kernel uses v_exp_f32 after a multiply by log2(e) and one reciprocal per column:
    its relative error grows with |logit|.  Large embedding weights make the logits span tens of units (PEAKED attention with
    large negative arguments, which the near-uniform random fixtures never exercise): clip form (T > 1, softmax launch /
    fused partial logits) and step form (T = 1, the per-frame fused kernel) must stay within 1e-4 of the oracle's exact exp."""
    A = A_KIN if v == 18 else pkg.ntu_graph().A
    m = pkg.AdaptiveGraphConvolution(ci, co, A).eval()
    _randomise(m, 90 + ci + t)
    with torch.no_grad():
        for name, prm in m.named_parameters():
            if ("a_conv" in name or "b_conv" in name) and name.endswith("weight"):
                prm.mul_(scale)
    sd = {k: v_.clone() for k, v_ in m.state_dict().items()}
    x = torch.rand(2, ci, t, v, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():                                    # the logits the softmax sees (models/a_gcn/a_gcn.py:53-62)
        a1 = torch.nn.functional.conv2d(x, sd["a_conv.0.weight"], sd["a_conv.0.bias"]).permute(0, 3, 1, 2).reshape(2, v, -1)
        a2 = torch.nn.functional.conv2d(x, sd["b_conv.0.weight"], sd["b_conv.0.bias"]).reshape(2, -1, v)
        logits = torch.matmul(a1, a2) / a1.shape[-1]
    assert float(logits.max() - logits.min()) > 20.0, float(logits.max() - logits.min())     # tens of units: a peaked softmax
    want = unit_scale_(m, sd, lambda s: o.adaptive_graph_conv(x, s), GCN_OUT_KEYS)
    got = m.to(DEV)(x.to(DEV)).cpu()
    check_parity(got, want, shape=(ci, co, t, v, scale), note="peaked attention (v_exp_f32 softmax)")


@pytest.mark.parametrize("v,t", [(25, 23), (18, 30)])
def test_fused_attention_entry_clip_form_equals_two_launch_route(v, t):
    """csk_agcn_embed_attention_f32, per-segment form, called directly (the module uses it at V = 18 only: at V = 25 it measured
    slower than the two launches) against csk_conv1x1_f32 + csk_agcn_attention_f32 on the same operands: the adjacencies agree
    to fp32 summation-order level."""
    from continual_skeletons_amd import native
    A = A_KIN if v == 18 else pkg.ntu_graph().A
    ci, co, n = 64, 128, 3
    m = pkg.AdaptiveGraphConvolution(ci, co, A).eval()
    _randomise(m, 31)
    m = m.to(DEV)
    ops = m._packed_ops(torch.device(DEV))
    x = torch.rand(n, ci, t, v, generator=torch.Generator().manual_seed(8)).to(DEV)
    inter, e_ch = m.inter_c, 6 * m.inter_c
    E = torch.empty((n * e_ch * t * v + 4,), device=DEV)[: n * e_ch * t * v].view(n, e_ch, t, v)
    native.check(native.lib().csk_conv1x1_f32(native.ptr(x), native.ptr(E), native.ptr(ops["w_embed"]), native.ptr(ops["b_embed"]), n, ci,
                                              e_ch, t, v, ci * t * v, t * v, e_ch * t * v, t * v, native.stream_of(x)), "conv1x1")
    want = m._attention(E, ops, n, t, v, e_ch * t * v, t * v)
    got = torch.empty((n, 3, v, v), device=DEV)
    ft = 128 // v
    scratch = torch.empty((n, 3, (t + ft - 1) // ft, v, v), device=DEV)
    native.check(native.lib().csk_agcn_embed_attention_f32(
        native.ptr(x), native.ptr(ops["w_embed_pairs"]), native.ptr(ops["b_embed_pairs"]), native.ptr(ops["a_sum"]), native.ptr(got),
        native.ptr(scratch), n, ci, inter, t, v, 0, ci * t * v, t * v, native.stream_of(x)), "embed_attention")
    check_parity(got.cpu(), want.cpu(), tol=1e-5, shape=(ci, co, t, v))


def test_adaptive_graph_conv_other_joint_counts():
    """A skeleton layout that is neither Kinetics nor NTU (V = 20, a random 3-subset adjacency): the dense kernels of gcn.hip
    (gcn_stage_dense_kernel for 128-row tiles, gcn_stage_kernel for 64-row ones) instead of gcn_dense.hip."""
    g = torch.Generator().manual_seed(21)
    A = (torch.rand(3, 20, 20, generator=g) < 0.2).float() * torch.rand(3, 20, 20, generator=g)
    for ci, co, t in ((128, 128, 9), (32, 64, 13)):
        m = pkg.AdaptiveGraphConvolution(ci, co, A).eval()
        _randomise(m, 5 + ci)
        sd = {k: v_.clone() for k, v_ in m.state_dict().items()}
        x = torch.rand(2, ci, t, 20, generator=torch.Generator().manual_seed(6))
        want = unit_scale_(m, sd, lambda s: o.adaptive_graph_conv(x, s), GCN_OUT_KEYS)
        check_parity(m.to(DEV)(x.to(DEV)).cpu(), want, shape=(ci, co, t, 20))


@pytest.mark.parametrize("graph,c", [("kinetics", 8), ("kinetics", 128), ("ntu", 8), ("ntu", 64)])
def test_agcn_block_clip_and_continual(graph, c):
    """SpatioTemporalBlock(GraphConv=AdaptiveGraphConvolution) in clip mode, and the CoAGCN block stepping
    (per-frame attention) against the oracle's continual block built on adaptive_graph_conv.  (c = 64: the fused embedding +
    attention launch on the Kinetics graph; the NTU graph keeps the two-launch route and gcn_stage_dense2_kernel<.., 25>.)"""
    A_KIN = pkg.kinetics_graph().A if graph == "kinetics" else pkg.ntu_graph().A
    blk = pkg.SpatioTemporalBlock(c, c, A_KIN, GraphConv=pkg.AdaptiveGraphConvolution).eval()
    _randomise(blk, 3)
    sd = {k: v.clone() for k, v in blk.state_dict().items()}
    x = torch.rand(2, c, 24, A_KIN.shape[-1], generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        want = o.st_block(x, sd, "", 1, True, gcn=o.adaptive_graph_conv)
    check_parity(blk.to(DEV)(x.to(DEV)).cpu(), want)
    co = pkg.CoSpatioTemporalBlock(c, c, A_KIN, padding=4, CoGraphConv=pkg.CoAdaptiveGraphConvolution).eval()
    co.load_state_dict(sd, strict=True)
    co = co.to(DEV)
    orc = o.CoBlockOracle(sd, "", 1, True, padding=4, gcn=o.adaptive_graph_conv)
    with torch.no_grad():
        for t in range(x.shape[2]):
            w = orc.forward_step(x[:, :, t])
            g = co.forward_step(x[:, :, t].contiguous().to(DEV))
            assert (w is None) == (g is None)
            if w is not None:
                check_parity(g.cpu(), w, note=t)


@pytest.mark.parametrize("native_plan,graph", [(True, "kinetics"), (False, "kinetics"), (True, "ntu")])
def test_coagcn_model_steps_vs_oracle(native_plan, graph):
    """The whole CoAGCN stack stepping against the oracle: through the native step executor (csk_co_plan with the adaptive
    graph-conv operands) and with every launch driven from Python -- the two must also agree bit for bit."""
    A = A_KIN if graph == "kinetics" else pkg.ntu_graph().A
    V = A.shape[-1]
    net = pkg.CoAGcn(A, input_shape=(3, 300, V, 2), num_classes=400 if V == 18 else 60, pool_size=4, pool_padding=1).eval()
    net.use_native_plan = native_plan
    randomise_unit_(net, 11, attn_scale=1 / V)          # O(1) activations through all ten blocks
    net_sd = net.state_dict()
    sd = {k.replace("0.1.", "").replace("0.0.residual", "residual"): v.clone() for k, v in net_sd.items()}
    x = torch.rand(1, 3, 96, V, 2, generator=torch.Generator().manual_seed(4))
    orc = o.CoStGcnOracle(sd, pool_size=4, pool_padding=1)
    for b in orc.blocks:
        b.gcn = o.adaptive_graph_conv
    want = []
    with torch.no_grad():
        for t in range(x.shape[2]):
            r = orc.forward_step(x[:, :, t])
            if r is not None:
                want.append(r)
    net = net.to(DEV)
    got = net.forward_steps(x.to(DEV)).cpu()
    assert bool(net.__dict__.get("_plan")) == native_plan
    want = torch.stack(want, dim=2)
    assert got.shape == want.shape and got.shape[2] >= 2
    check_parity(got, want)
    if native_plan:                                     # same kernels, same launches: the Python engine gives the same bits
        net.use_native_plan = False
        net._n = None                                   # re-bind (clean state, no plan)
        assert torch.equal(net.forward_steps(x.to(DEV)).cpu(), got) and not net.__dict__.get("_plan")
        net.use_native_plan = True
        net._n = None
        net.forward_steps(x.to(DEV))
    net.clean_state()                                   # the same frames in 4-frame cycles (multi-slot GCN stage)
    xd = x.to(DEV)
    cyc = []
    for t in range(0, x.shape[2], 4):
        cyc += net.forward_cycle([xd[:, :, t + f].contiguous() for f in range(4)])
    assert len(cyc) == got.shape[2] and all(torch.equal(c.cpu(), got[:, :, i]) for i, c in enumerate(cyc))


def test_full_agcn_golden():
    """Fixture G8: the reference's whole AGcn model (Kinetics shape, closed-form weights with non-trivial a_conv /
    b_conv) -- logits and the layer 1 / 5 / 8 / 10 activations of the HIP path against the reference's outputs."""
    a, sd, x = g8_state_dict()
    net = pkg.AGcn(A_KIN, (3, 300, 18, 2), 400).eval()
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV)
    taps = {}
    hooks = [net.layers[f"layer{i}"].register_forward_hook(lambda mod, inp, out, i=i: taps.__setitem__(i, out))
             for i in (1, 5, 8, 10)]
    logits = net(x.to(DEV))
    for h in hooks:
        h.remove()
    check_parity(logits.cpu(), a["logits"])
    for i in (1, 5, 8, 10):
        check_parity(taps[i].cpu().reshape(-1)[::997], a[f"layer{i}_sub"])


def test_agcn_clip_latency_mode_golden():
    """AGcn.set_latency_mode: the temporal convs split their K loop (csk_tcn_stage_splitk_f32, V = 18), the adaptive graph
    convs keep their kernels (no split form): fixture G8 within 1e-4, default mode restored bit for bit."""
    a, sd, x = g8_state_dict()
    net = pkg.AGcn(A_KIN, (3, 300, 18, 2), 400).eval()
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV)
    base = net(x.to(DEV)).cpu()
    net.set_latency_mode(4)
    assert net.layers.layer6.clip_split_k == 4 and net.layers.layer6.gcn._clip_ksplit() == 1
    lat = net(x.to(DEV)).cpu()
    check_parity(lat, a["logits"], note="A-GCN latency mode vs G8")
    check_parity(lat, base, note="A-GCN latency mode vs default (summation order)")
    net.set_latency_mode(0)
    assert torch.equal(net(x.to(DEV)).cpu(), base)
