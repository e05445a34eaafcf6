"""GPU parity, clip path: HIP stage kernels (through the C ABI) vs golden vectors from the reference and
vs the CPU oracle on seeded inputs.  Tolerance: |hip - ref| <= 1e-4 absolute on O(1) activations
(BASELINE.json north_star: "within 1e-4 fp32")."""
import numpy as np
import pytest
import torch

import _bootstrap
from oracle import stgcn_oracle as o
from tests.helpers import BLOCK_OUT_KEYS, check_parity, g6_state_dict, load_golden, max_err, unit_scale_

pytestmark = pytest.mark.gpu
TOL = 1e-4
pkg = _bootstrap.load()
DEV = "cuda:0"


def _A(v=25):
    return (pkg.ntu_graph() if v == 25 else pkg.kinetics_graph()).A


@pytest.mark.parametrize("tag,ci,co", [("eq", 4, 4), ("neq", 3, 8)])
def test_graph_conv_golden(tag, ci, co):
    a, sd = load_golden(f"g1_gcn_{tag}")
    m = pkg.GraphConvolution(ci, co, _A()).eval()
    m.load_state_dict(sd, strict=True)
    y = m.to(DEV)(torch.from_numpy(a["x"]).to(DEV))
    check_parity(y.cpu(), a["y"])


@pytest.mark.parametrize("tag", ["k9s1p4", "k9s2p4", "k1s2p0", "k9s1p0"])
def test_temporal_conv_golden(tag):
    a, sd = load_golden(f"g2_tcn_{tag}")
    k, s, p = (int(v) for v in a["meta"])
    m = pkg.TemporalConvolution(4, 4, k, s, p).eval()
    m.load_state_dict(sd, strict=True)
    y = m.to(DEV)(torch.from_numpy(a["x"]).to(DEV))
    assert tuple(y.shape) == a["y"].shape
    check_parity(y.cpu(), a["y"])


@pytest.mark.parametrize("tag", ["nores", "ident", "convres", "strided", "nopad", "nopad_strided"])
def test_block_golden(tag):
    a, sd = load_golden(f"g3_block_{tag}")
    ci, co, s, res, tp = (int(v) for v in a["meta"])
    m = pkg.SpatioTemporalBlock(ci, co, _A(), s, bool(res), temporal_padding=tp).eval()
    m.load_state_dict(sd, strict=True)
    y = m.to(DEV)(torch.from_numpy(a["x"]).to(DEV))
    assert tuple(y.shape) == a["y"].shape
    check_parity(y.cpu(), a["y"])


def test_stack_golden():
    a, sd = load_golden("g4_stack")
    A = _A()
    stack = torch.nn.Sequential(
        pkg.SpatioTemporalBlock(3, 3, A, residual=False), pkg.SpatioTemporalBlock(3, 3, A),
        pkg.SpatioTemporalBlock(3, 4, A, stride=2),
    ).eval()
    stack.load_state_dict(sd, strict=True)
    y = stack.to(DEV)(torch.from_numpy(a["x"]).to(DEV))
    check_parity(y.cpu(), a["y"])


def test_config1_block_golden():
    """BASELINE config 1: SpatioTemporalBlock(3, 64, A_ntu, residual=False) on (2, 3, 300, 25)."""
    from closed_form import closed_form_input

    a, sd = load_golden("g5_config1_block")
    m = pkg.SpatioTemporalBlock(3, 64, _A(), residual=False).eval()
    m.load_state_dict(sd, strict=True)
    x = torch.from_numpy(closed_form_input((2, 3, 300, 25), salt=5.0)).to(DEV)
    y = m.to(DEV)(x).cpu()
    assert tuple(y.shape) == tuple(a["y_shape"])
    check_parity(y.reshape(-1)[::7], a["y_sub7"])
    assert np.allclose(y.sum(dim=(0, 2, 3)).numpy(), a["y_chan_sum"], rtol=1e-4)


@pytest.mark.parametrize("tag", ["ntu", "kin"])
def test_full_stgcn_golden(tag):
    a, sd, x = g6_state_dict(tag)
    v, classes = (25, 60) if tag == "ntu" else (18, 400)
    net = pkg.StGcn(_A(v), input_shape=(3, 300, v, 2), num_classes=classes).eval()
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV)
    taps = {}
    hooks = [net.layers[f"layer{i}"].register_forward_hook(lambda m, inp, out, i=i: taps.__setitem__(i, out))
             for i in (1, 5, 8, 10)]
    logits = net(x.to(DEV)).cpu()
    for h in hooks:
        h.remove()
    for i in (1, 5, 8, 10):
        check_parity(taps[i].cpu().reshape(-1)[::997], a[f"layer{i}_sub"], note=f"layer{i}")
    check_parity(logits, a["logits"])


@pytest.mark.parametrize("ci,co,stride,res,T,N", [
    (64, 64, 1, True, 37, 3),      # identity residual, ragged last tile
    (64, 128, 2, True, 50, 2),     # strided conv residual, MT=128 path
    (128, 256, 2, True, 31, 2),    # odd T with stride 2, two M tiles
    (3, 64, 1, False, 300, 1),     # layer-1 shape
    (20, 12, 1, True, 9, 1),       # channel counts that are not multiples of the tile sizes
])
def test_block_vs_oracle_seeded(ci, co, stride, res, T, N):
    torch.manual_seed(1234 + ci + co + T)
    m = pkg.SpatioTemporalBlock(ci, co, _A(), stride, res).eval()
    with torch.no_grad():
        for name, prm in m.named_parameters():
            if name.endswith("graph_attn") or name.endswith("bn.weight") or name.endswith("residual.1.weight"):
                prm.copy_(torch.rand_like(prm) + 0.5)
            elif name.endswith("bias"):
                prm.copy_(torch.rand_like(prm) - 0.5)
        for name, buf in m.named_buffers():
            if name.endswith("running_var"):
                buf.copy_(torch.rand_like(buf) + 0.5)
            elif name.endswith("running_mean"):
                buf.copy_(torch.rand_like(buf) - 0.5)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.rand(N, ci, T, 25)
    want = unit_scale_(m, sd, lambda s: o.st_block(x, s, "", stride, res), BLOCK_OUT_KEYS)
    got = m.to(DEV)(x.to(DEV)).cpu()
    check_parity(got, want, shape=(ci, co, stride, res, T, N))


def test_errors_are_loud():
    m = pkg.SpatioTemporalBlock(4, 4, _A())
    with pytest.raises(RuntimeError, match="inference-only"):
        m.to(DEV)(torch.rand(1, 4, 20, 25, device=DEV))
    m.eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.rand(1, 4, 20, 25))
    with pytest.raises(RuntimeError):
        m.to(DEV)(torch.rand(1, 4, 20, 25, device=DEV).double())


def test_general_gcn_kernel_matches_sparse_fast_path():
    """The dense-capable GCN kernel (used for A-GCN / arbitrary adjacencies) and the sparse-graph fast path must
    agree on a skeleton graph; run the general one in a subprocess (diagnostic switch is read at library load)."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, torch; sys.path.insert(0, %r); import _bootstrap; pkg = _bootstrap.load();"
        "torch.manual_seed(3); m = pkg.GraphConvolution(64, 128, pkg.ntu_graph().A).eval();"
        "[p.data.copy_(torch.rand_like(p) + 0.5) for n, p in m.named_parameters() if n.endswith('graph_attn') or 'bn' in n and n.endswith('weight')];"
        "x = torch.rand(2, 64, 33, 25); y = m.to('cuda:0')(x.to('cuda:0')).cpu(); torch.save(y, sys.argv[1])"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for general in (False, True):
        path = f"/tmp/gcn_{int(general)}.pt"
        env = dict(os.environ)
        env.pop("CSK_DIAG", None)
        if general:
            env.update(CSK_DIAG="1", CSK_GCN_GENERAL="1")
        subprocess.check_call([sys.executable, "-c", code, path], env=env)
        outs.append(torch.load(path))
    check_parity(outs[0], outs[1], tol=1e-5, note="general vs sparse GCN kernel")


def test_tile_families_are_bitwise_interchangeable():
    """The 16x16x4 tile family (csrc/tcn16.hip, csrc/step16.hip:gcn16_kernel) stands in for the 32x32x2 kernels where a launch
    shape favours it, on the claim that both walk the K loop in the same (chunk, tap / subset, channel) order and that
    v_mfma_f32_16x16x4_f32 is the same fmaf chain: stride-1 blocks (identity and conv residual, 64 and 128 channels) and a
    step-layout graph conv must come out BIT FOR BIT the same with the family forced on everywhere (CSK_TCN16=2, CSK_GCN16=2)
    and switched off (=1).  (The stride-2 temporal conv of the family walks 4-channel chunks -- another fp32 order -- and is
    compared with the oracle like every kernel.)  Subprocesses: the switches are read when the library is loaded."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, torch; sys.path.insert(0, %r); import _bootstrap, bench; pkg = _bootstrap.load(); A = pkg.ntu_graph().A; outs = [];\n"
        "for (ci, co, res) in [(64, 64, True), (128, 128, True), (8, 64, True), (64, 64, False)]:\n"
        "    b = pkg.SpatioTemporalBlock(ci, co, A, stride=1, residual=res).eval(); bench.randomise_(b, 3); b = b.to('cuda:0')\n"
        "    x = torch.rand((3, ci, 45, 25), generator=torch.Generator().manual_seed(5)).to('cuda:0'); outs.append(b(x).cpu())\n"
        "g = pkg.GraphConvolution(64, 64, A).eval(); bench.randomise_(g, 4); g = g.to('cuda:0')\n"
        "P = 16 * 25; xs = torch.rand((4, 64, P), generator=torch.Generator().manual_seed(6)).to('cuda:0'); ys = torch.empty((4, 64, P), device='cuda:0')\n"
        "g.stage(xs, ys, n_seg=4, frames=16, x_strides=(64 * P, P), y_strides=(64 * P, P)); outs.append(ys.cpu()); torch.save(outs, sys.argv[1])\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for mode in ("2", "1"):
        path = f"/tmp/tile16_{mode}.pt"
        env = dict(os.environ, CSK_DIAG="1", CSK_TCN16=mode, CSK_GCN16=mode)
        subprocess.check_call([sys.executable, "-c", code, path], env=env)
        res.append(torch.load(path))
    assert len(res[0]) == len(res[1]) == 5
    for a_, b_ in zip(res[0], res[1]):
        assert torch.equal(a_, b_) and bool(torch.isfinite(a_).all())


def test_clip_forward_is_graph_capturable():
    """include/cskel.h promises launches without allocation or synchronisation: a whole 10-block forward must be
    capturable into a hipGraph and replay bit-identically."""
    a, sd, x = g6_state_dict("ntu")
    net = pkg.StGcn(_A()).eval()
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV)
    xd = x[:1].to(DEV)
    for _ in range(2):
        ref = net(xd)                     # warm-up: folds weights, raises the LDS caps, fills the caching allocator
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = net(xd)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    xd.copy_(x[1:2].to(DEV))              # new input in the captured buffer -> replay computes the new result
    g.replay()
    torch.cuda.synchronize()
    check_parity(out.cpu(), a["logits"][1:2])


@pytest.mark.parametrize("ci,co,v,t,n,ksplit", [(64, 64, 25, 9, 3, 4), (64, 128, 25, 5, 2, 8), (256, 256, 25, 2, 2, 32), (128, 256, 18, 3, 5, 6),
                                                (40, 70, 25, 4, 2, 3)])
def test_graph_conv_split_k_entry_vs_oracle_and_unsplit(ci, co, v, t, n, ksplit):
    """csk_gcn_stage_splitk_f32 (latency mode: the K loop of the graph conv cut into channel ranges over workgroups +
    gcn_reduce_kernel) on clip-layout operands: within 1e-4 of the oracle and of csk_gcn_stage_f32 (summation order only),
    identity and conv gcn_residual, ragged channel counts, both skeleton graphs, more ranges asked for than channels allow."""
    from continual_skeletons_amd import native
    g = torch.Generator().manual_seed(ci + co + ksplit)
    m = pkg.GraphConvolution(ci, co, (pkg.ntu_graph() if v == 25 else pkg.kinetics_graph()).A).eval()
    with torch.no_grad():
        for name, prm in m.named_parameters():
            if name.endswith("graph_attn") or ("bn" in name and name.endswith("weight")) or name.endswith("gcn_residual.1.weight"):
                prm.copy_(torch.rand(prm.shape, generator=g) + 0.5)
            elif name.endswith("bias"):
                prm.copy_(torch.rand(prm.shape, generator=g) - 0.5)
            elif name.endswith("weight") and prm.dim() == 4:
                prm.copy_(torch.randn(prm.shape, generator=g) * (0.5 / ci) ** 0.5)
        for name, buf in m.named_buffers():
            if name.endswith("running_var"):
                buf.copy_(torch.rand(buf.shape, generator=g) + 0.5)
            elif name.endswith("running_mean"):
                buf.copy_(torch.rand(buf.shape, generator=g) - 0.5)
    sd = {k: x.clone() for k, x in m.state_dict().items()}
    x = torch.rand(n, ci, t, v, generator=g)
    with torch.no_grad():
        want = o.graph_conv(x, sd, "")
    m = m.to(DEV)
    xd = x.to(DEV)
    plain = m(xd)
    ops = m._packed_ops(xd.device)
    y = torch.empty_like(plain)
    part = torch.full((n * ksplit, co, t * v), float("nan"), device=DEV)          # every element the reduction reads must be written
    rc = native.lib().csk_gcn_stage_splitk_f32(
        native.ptr(xd), native.ptr(y), native.ptr(ops["w"]), native.ptr(ops["bias"]), native.ptr(ops["ell_src"]),
        native.ptr(ops["ell_val"]), native.ptr(ops["ell_cnt_host"]), ops["ell_w"], n, ci, co, t, v, ci * t * v, t * v, co * t * v, t * v,
        ops["res_mode"], ksplit, native.ptr(part), native.stream_of(xd))
    native.check(rc, "csk_gcn_stage_splitk_f32")
    check_parity(y.cpu(), want, shape=(ci, co, v, t, n, ksplit))
    check_parity(y.cpu(), plain.cpu(), note="split-K vs unsplit graph conv (summation order)")


@pytest.mark.parametrize("ci,co,stride,res,T,N,ksplit", [
    (64, 64, 1, True, 37, 2, 4),       # identity residual, 64-row tiles, ragged last tile
    (64, 128, 2, True, 50, 2, 8),      # strided conv residual (rides in split 0), 128-row tiles
    (128, 256, 2, True, 31, 1, 5),     # a factor that does not divide the 16 chunks evenly
    (256, 256, 1, True, 12, 2, 32),    # as many ranges as 8-channel chunks
    (64, 64, 1, False, 20, 1, 3),      # no residual
    (40, 72, 1, True, 11, 2, 6),       # ragged channel counts: more ranges asked for than the 5 chunks allow
])
def test_temporal_conv_split_k_entry_vs_oracle_and_unsplit(ci, co, stride, res, T, N, ksplit):
    """csk_tcn_stage_splitk_f32 (clip latency mode: the K loop of the 9 x 1 temporal conv cut into channel ranges over
    workgroups + tcn_reduce_kernel): the block's output within 1e-4 of the oracle and of the unsplit launch (summation
    order only), for every residual form; unwritten partial sums would show up as NaN."""
    torch.manual_seed(77 + ci + co + T)
    m = pkg.SpatioTemporalBlock(ci, co, _A(), stride, res).eval()
    with torch.no_grad():
        for name, prm in m.named_parameters():
            if name.endswith("graph_attn") or name.endswith("bn.weight") or name.endswith("residual.1.weight"):
                prm.copy_(torch.rand_like(prm) + 0.5)
            elif name.endswith("bias"):
                prm.copy_(torch.rand_like(prm) - 0.5)
        for name, buf in m.named_buffers():
            if name.endswith("running_var"):
                buf.copy_(torch.rand_like(buf) + 0.5)
            elif name.endswith("running_mean"):
                buf.copy_(torch.rand_like(buf) - 0.5)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.rand(N, ci, T, 25)
    want = unit_scale_(m, sd, lambda s: o.st_block(x, s, "", stride, res), BLOCK_OUT_KEYS)
    m = m.to(DEV)
    plain = m(x.to(DEV)).cpu()
    from continual_skeletons_amd import blocks
    m.clip_split_k = ksplit                                   # the temporal conv only (the graph conv has its own test)
    # poison the partial-sum scratch AFTER sizing it for this launch (the block's own SplitScratch hands the same buffer
    # back to the forward below): a partial sum that is read without having been written shows up as NaN
    t_out = (T + 2 * 4 - 9) // stride + 1
    blocks._scratch_of(m).get(torch.device(DEV), ksplit * N * co * t_out * 25).fill_(float("nan"))
    got = m(x.to(DEV)).cpu()
    check_parity(got, want, shape=(ci, co, stride, res, T, N, ksplit))
    check_parity(got, plain, note="split-K vs unsplit temporal conv (summation order)")


def test_clip_latency_mode_full_model_golden_and_batch_invariance():
    """StGcn.set_latency_mode (split-K on every block's graph conv and temporal conv): G6 logits and layer taps within
    1e-4; a clip's logits are BITWISE the same alone and in a batch (the split factor never depends on the batch); the
    default mode comes back with split_k = 0."""
    a, sd, x = g6_state_dict("ntu")
    net = pkg.StGcn(_A()).eval()
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV)
    base = net(x.to(DEV)).cpu()
    net.set_latency_mode(4)
    assert all(net.layers[f"layer{i}"].clip_split_k == 4 for i in range(1, 11)) and net.layers.layer2.gcn.clip_split_k == 4
    taps = {}
    hooks = [net.layers[f"layer{i}"].register_forward_hook(lambda m, inp, out, i=i: taps.__setitem__(i, out)) for i in (1, 5, 8, 10)]
    both = net(x.to(DEV)).cpu()
    for h in hooks:
        h.remove()
    for i in (1, 5, 8, 10):
        check_parity(taps[i].cpu().reshape(-1)[::997], a[f"layer{i}_sub"], note=f"latency mode layer{i}")
    check_parity(both, a["logits"], note="latency mode logits")
    check_parity(both, base, note="latency mode vs default (summation order)")
    one = net(x[:1].to(DEV)).cpu()
    assert torch.equal(one, both[:1])                         # batch-invariant for a fixed split_k
    # above max_sequences (default 6 = batch 3 at M = 2) the mode runs the default kernels: never slower than the default mode,
    # bitwise the default mode's logits
    x4 = torch.cat([x, x]).to(DEV)
    big = net(x4).cpu()
    net.set_latency_mode(4, max_sequences=8)                  # ... a bound that takes batch 4 in
    big_split = net(x4).cpu()
    assert torch.equal(big_split[:2], both) and torch.equal(big_split[2:], both)
    net.set_latency_mode(0)
    assert torch.equal(net(x.to(DEV)).cpu(), base)
    assert torch.equal(net(x4).cpu(), big) and torch.equal(big[:2], base)
    with pytest.raises(ValueError):
        pkg.set_clip_latency_mode(torch.nn.Linear(2, 2))


@pytest.mark.parametrize("v,t,cin,co", [(25, 52, 3, 64), (25, 16, 3, 64), (25, 300, 3, 64), (18, 37, 3, 64), (25, 20, 2, 64), (18, 33, 4, 64),
                                        (25, 7, 3, 64), (25, 40, 3, 128), (25, 24, 1, 8)])
def test_first_block_as_one_launch_is_bitwise_the_two_launches(v, t, cin, co):
    """csk_block_few_channels_f32 (layer 1 of the stacks, st_gcn.py:30: graph conv formed on the fly inside the temporal conv's
    tile) against csk_gcn_stage_f32 + csk_tcn_stage_f32: bit for bit -- frame counts that are not whole 16-frame tiles, both
    skeletons, 1-4 input channels, more than one m-tile; the input between NaN guards in the fused run."""
    import bench
    A_ = (pkg.ntu_graph() if v == 25 else pkg.kinetics_graph()).A
    blk = pkg.SpatioTemporalBlock(cin, co, A_, residual=False).eval()
    bench.randomise_(blk, cin + t)
    blk = blk.to(DEV)
    x = torch.rand((3, cin, t, v), generator=torch.Generator().manual_seed(t)).to(DEV)
    buf = torch.full((x.numel() + 8192,), float("nan"), device=DEV)
    xg = buf[4096: 4096 + x.numel()].view(x.shape)
    xg.copy_(x)
    blk.fuse_few_channels = True                              # opt-in (measured neutral in time: blocks.py)
    assert blk._few_channels_fusable(xg)
    one = blk(xg).cpu()
    blk.fuse_few_channels = False
    two = blk(x).cpu()
    assert bool(torch.isfinite(one).all())
    assert torch.equal(one, two)
