"""CPU tests of the host layer: C-ABI symbols, state_dict layouts, folding arithmetic, loud failures."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import _bootstrap
from oracle import stgcn_oracle as o
from tests.helpers import load_golden

pkg = _bootstrap.load()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "cskel.h")).read()
    declared = set(re.findall(r"\b(csk_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed from include/cskel.h"
    lib = ctypes.CDLL(pkg.native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in cskel.h but not exported"
    assert declared - {"csk_last_error"} == set(pkg.native.SIGNATURES), "ctypes table out of sync with cskel.h"
    assert pkg.native.lib().csk_abi_version() == pkg.native.ABI_VERSION == 15


def test_argument_errors_do_not_need_a_gpu():
    lib = pkg.native.lib()
    rc = lib.csk_tcn_stage_f32(None, None, None, None, None, None, 1, 1, 1, 1, 25, 9, 1, 4, 0, 0, 0, 0, 1, None)
    assert rc == -1 and b"null pointer" in lib.csk_last_error()
    with pytest.raises(RuntimeError, match="null pointer"):
        pkg.native.check(rc, "csk_tcn_stage_f32")


def test_graph_matches_reference_fixture():
    a, _ = load_golden("g0_graphs")
    assert np.array_equal(pkg.ntu_graph().A, a["ntu"]) and np.array_equal(pkg.kinetics_graph().A, a["kinetics"])


def test_state_dict_layouts_match_reference():
    A = pkg.ntu_graph().A
    _, sd = load_golden("g3_block_strided")           # dumped from the reference's SpatioTemporalBlock
    blk = pkg.SpatioTemporalBlock(2, 4, A, stride=2)
    assert list(blk.state_dict().keys()) == list(sd.keys())
    assert all(blk.state_dict()[k].shape == v.shape for k, v in sd.items())
    blk.load_state_dict(sd, strict=True)
    # continual container layouts (tests/test_cost_gcn.py:97-98,145,193-198)
    co = pkg.CoSpatioTemporalBlock(2, 4, A, stride=2, padding=4)
    keys = set(co.state_dict().keys())
    assert {"0.0.residual.t_conv.weight", "0.1.gcn.g_conv.0.weight", "0.1.tcn.bn.running_var"} <= keys
    assert not any(k.startswith(("gcn.", "tcn.", "residual.")) for k in keys)
    mapping = {"res": "0.0.", "gcn": "0.1.", "tcn": "0.1."}
    co.load_state_dict({mapping[k[:3]] + k: v for k, v in sd.items()}, strict=True)    # reference Co layout
    co.load_state_dict(sd, strict=True)                                                # plain layout
    ident = pkg.CoSpatioTemporalBlock(4, 4, A, padding=4)
    assert all(k.startswith("0.1.") for k in ident.state_dict())
    nores = pkg.CoSpatioTemporalBlock(4, 4, A, residual=False, padding=4)
    assert all(k.startswith(("gcn.", "tcn.")) for k in nores.state_dict())
    assert (co.delay, co.receptive_field, co.stride, co.padding) == (4, 9, 2, 4)
    assert pkg.CoSpatioTemporalBlock(4, 4, A).delay == 8                               # padding=0 default


def test_costgcn_keys_and_map_state_dict():
    A = pkg.ntu_graph().A
    reg, co = pkg.StGcn(A), pkg.CoStGcn(A)
    assert "layers.layer2.0.1.gcn.g_conv.0.weight" in co.state_dict()
    assert "layers.layer8.0.0.residual.t_conv.weight" in co.state_dict()
    assert "layers.layer1.gcn.bn.weight" in co.state_dict()
    mapped = co.map_state_dict(reg.state_dict())
    assert set(mapped) == set(co.state_dict())
    co.load_state_dict(mapped, strict=True)
    co.load_state_dict(reg.state_dict(), strict=True)      # regular layout loads directly as well
    assert (co.receptive_field, co.padding, co.stride, co.pool_size, co.pool_padding) == (153, 76, 4, 75, 19)


def test_costgcn_key_map_matches_the_reference_fixture():
    """G9 (tests/golden/make_golden.py): the continual key layout and the output of the reference's own
    CoModelBase.map_state_dict (models/base.py:200-224, executed unbound on a stub with that layout) on the reference
    StGcn's keys.  CoStGcn must expose the same keys in the same order and map a regular state dict the same way:
    strict, non-strict (unknown keys dropped) and already-continual (returned as is)."""
    import numpy as np
    d = np.load(os.path.join(ROOT, "tests", "golden", "g9_key_map.npz"))
    regular, co_keys = [str(k) for k in d["regular_keys"]], [str(k) for k in d["co_keys"]]
    net = pkg.CoStGcn(pkg.ntu_graph().A)
    assert list(net.state_dict().keys()) == co_keys
    assert list(pkg.StGcn(pkg.ntu_graph().A).state_dict().keys()) == regular
    sd = {k: i for i, k in enumerate(regular)}
    strict = net.map_state_dict(dict(sd), strict=True)
    assert list(strict.keys()) == [str(k) for k in d["mapped_strict_keys"]] and list(strict.values()) == list(d["mapped_strict_pos"])
    loose = net.map_state_dict(dict(sd, **{"not.a.key": -1}), strict=False)
    assert list(loose.keys()) == [str(k) for k in d["mapped_loose_keys"]] and list(loose.values()) == list(d["mapped_loose_pos"])
    with pytest.raises(KeyError):                                   # the reference raises on an unknown key when strict
        net.map_state_dict(dict(sd, **{"not.a.key": -1}), strict=True)
    same = net.map_state_dict({k: i for i, k in enumerate(co_keys)}, strict=True)
    assert list(same.keys()) == [str(k) for k in d["mapped_same_keys"]] == co_keys


def test_load_pretrained_pt_and_ckpt(tmp_path):
    """Weight import path: a regular ST-GCN state dict saved as .pt, or inside a Lightning-style
    .ckpt, loads into StGcn and -- through map_loaded_weights (models/base.py:226-227) -- into CoStGcn; a state
    dict already in the continual layout passes through unchanged; junk files are refused."""
    import torch
    A = pkg.ntu_graph().A
    src = pkg.StGcn(A)
    with torch.no_grad():
        for prm in src.parameters():
            prm.add_(0.01)
    pt, ckpt, junk = tmp_path / "w.pt", tmp_path / "w.ckpt", tmp_path / "junk.pt"
    torch.save(src.state_dict(), pt)
    torch.save({"state_dict": src.state_dict(), "epoch": 3}, ckpt)
    torch.save([1, 2, 3], junk)
    reg, co = pkg.StGcn(A), pkg.CoStGcn(A)
    pkg.load_pretrained(reg, str(pt))
    pkg.load_pretrained(co, str(ckpt))
    want = src.state_dict()
    assert all(torch.equal(v, want[k]) for k, v in reg.state_dict().items())
    short = lambda k: k.replace("0.1.", "").replace("0.0.residual", "residual")  # noqa: E731
    assert all(torch.equal(v, want[short(k)]) for k, v in co.state_dict().items())
    assert co.map_loaded_weights(str(pt), co.state_dict()) is not None
    same = co.state_dict()
    assert co.map_state_dict(same) is same                      # already in the continual layout: unchanged
    extra = dict(src.state_dict(), not_a_key=torch.zeros(1))
    torch.save(extra, pt)
    with pytest.raises((RuntimeError, KeyError)):
        pkg.load_pretrained(pkg.CoStGcn(A), str(pt))            # strict: unexpected key
    res = pkg.load_pretrained(pkg.CoStGcn(A), str(pt), strict=False)
    assert not res.missing_keys
    with pytest.raises(RuntimeError, match="not a state dict"):
        pkg.load_pretrained(reg, str(junk))


def test_fold_reproduces_oracle_pointwise():
    """Evaluate the PACKED operands with plain numpy (test-only) and compare with the oracle."""
    a, sd = load_golden("g3_block_strided")
    f = pkg.fold.fold_graph_conv(sd, "gcn.")
    x = torch.from_numpy(a["x"]).double()
    n, ci, t, v = x.shape
    a_eff = (sd["gcn.A"] * sd["gcn.graph_attn"]).double()
    # ELL round trip
    dense = torch.zeros(3, v, v, dtype=torch.float64)
    for i in range(3):
        for w in range(v):
            for e in range(int(f["ell_cnt_host"][i])):
                dense[i, int(f["ell_src"][i, w, e]), w] += float(f["ell_val"][i, w, e])
    assert torch.allclose(dense, a_eff.float().double())
    xa = [torch.einsum("nctv,vw->nctw", x, a_eff[i]) for i in range(3)] + [x]
    y = sum(torch.einsum("nctw,co->notw", xa[r], f["w"][r, :ci, : f["c_out"]].double()) for r in range(4))
    y = torch.relu(y + f["bias"][: f["c_out"]].double()[None, :, None, None])
    with torch.no_grad():
        want = o.graph_conv(torch.from_numpy(a["x"]), sd, "gcn.")
    assert float((y - want.double()).abs().max()) < 1e-5


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: no module of the product package may import or execute it."""
    import ast
    pkg_dir = os.path.join(ROOT, "continual-skeletons_amd")
    for fn in os.listdir(pkg_dir):
        if not fn.endswith(".py"):
            continue
        tree = ast.parse(open(os.path.join(pkg_dir, fn)).read())
        for node in ast.walk(tree):
            mods = []
            if isinstance(node, ast.Import):
                mods = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                mods = [node.module or ""]
            assert not any(m.split(".")[0] == "oracle" for m in mods), f"{fn} imports the oracle"


def _oracle_imports(path):
    """(enclosing top-level function or None, module) for every import of the oracle in a source file."""
    import ast
    tree = ast.parse(open(path).read())
    found = []
    for top in tree.body:
        owner = top.name if isinstance(top, (ast.FunctionDef, ast.ClassDef)) else None
        for node in ast.walk(top):
            mods = [a.name for a in node.names] if isinstance(node, ast.Import) else \
                   [node.module or ""] if isinstance(node, ast.ImportFrom) else []
            found += [(owner, m) for m in mods if m.split(".")[0] == "oracle"]
    return found


def test_oracle_is_only_used_as_checker_or_cpu_baseline():
    """Outside tests/ the oracle may be touched by bench.py's cpu_baseline leg and __graft_entry__.smoke() only;
    the tools never use it."""
    assert {o for o, _ in _oracle_imports(os.path.join(ROOT, "bench.py"))} <= {"cpu_baseline_clip", "cpu_baseline_step"}
    assert {o for o, _ in _oracle_imports(os.path.join(ROOT, "__graft_entry__.py"))} <= {"smoke", "build"}
    tools = os.path.join(ROOT, "tools")
    for fn in os.listdir(tools):
        if fn.endswith(".py"):
            assert not _oracle_imports(os.path.join(tools, fn)), f"tools/{fn} imports the oracle"
    assert not _oracle_imports(os.path.join(ROOT, "_bootstrap.py"))


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(pkg.native, "_lib", None)
    monkeypatch.setattr(pkg.native, "LIB_PATH", "/nonexistent/libcskel_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.native.lib()


class _ForeignHparams:                                       # stands for Ride's AttributeDict & co. in a Lightning checkpoint
    def __init__(self):
        self.learning_rate = 0.1


def test_lightning_style_checkpoint_with_hparams(tmp_path):
    """a real Lightning / Ride .ckpt carries `hyper_parameters` objects.  argparse.Namespace loads under the
    safe unpickler; any other class is refused with a message that says what to do; trusted=True loads it."""
    import argparse

    import torch
    A = pkg.ntu_graph().A
    src = pkg.StGcn(A)
    ns, foreign = tmp_path / "ns.ckpt", tmp_path / "foreign.ckpt"
    torch.save({"state_dict": src.state_dict(), "hyper_parameters": argparse.Namespace(graph="ntu_rgbd", lr=0.1)}, ns)
    torch.save({"state_dict": src.state_dict(), "hyper_parameters": _ForeignHparams()}, foreign)
    pkg.load_pretrained(pkg.CoStGcn(A), str(ns))
    with pytest.raises(RuntimeError, match="trusted=True"):
        pkg.load_pretrained(pkg.CoStGcn(A), str(foreign))
    pkg.load_pretrained(pkg.CoStGcn(A), str(foreign), trusted=True)


def test_fusion_file_loaders(tmp_path):
    """.npy prediction arrays and the pickled (names, labels) label file of scripts/multi_stream_eval.py:16-31."""
    import pickle

    import numpy as np
    from continual_skeletons_amd import fusion
    np.save(tmp_path / "p.npy", np.zeros((3, 5), dtype=np.float32))
    np.save(tmp_path / "bad.npy", np.zeros((3,), dtype=np.float32))
    (tmp_path / "l.pkl").write_bytes(pickle.dumps((["a", "b", "c"], [4, 0, 2]), protocol=2))
    assert fusion.load_preds(tmp_path / "p.npy").shape == (3, 5)
    assert fusion.load_labels(tmp_path / "l.pkl").tolist() == [4, 0, 2]
    # a label file is plain lists of str / int: anything that needs a global (here: os.system) is refused, not executed
    (tmp_path / "evil.pkl").write_bytes(b"cos\nsystem\n(S'true'\ntR.")
    with pytest.raises(pickle.UnpicklingError, match="refusing"):
        fusion.load_labels(tmp_path / "evil.pkl")
    (tmp_path / "np.pkl").write_bytes(pickle.dumps((["a"], np.array([3])), protocol=2))
    with pytest.raises(pickle.UnpicklingError):
        fusion.load_labels(tmp_path / "np.pkl")
    assert fusion.load_labels(tmp_path / "np.pkl", trusted=True).tolist() == [3]
    with pytest.raises(ValueError):
        fusion.load_preds(tmp_path / "bad.npy")
    with pytest.raises(ValueError):
        fusion.load_preds(tmp_path / "p.txt")
    with pytest.raises(FileNotFoundError):
        fusion.load_preds(tmp_path / "missing.npy")


def test_co_block_step_argument_errors_are_reported_without_a_gpu():
    """csk_co_block_step_f32 validates its arguments before it touches the device: bad geometry comes back as a
    negative return code with a message (no launch, so this runs on the CPU-only container)."""
    import ctypes as C
    lib = pkg.native.lib()
    fake = C.c_void_p(0x1000)                    # non-null, 16-byte aligned; never dereferenced on these paths
    cnt = (C.c_int32 * 3)(1, 1, 4)

    def call(c_in=64, c_out=64, y_slots=16, V=25, P=200, n_skel=8, res_mode=1, gcn_res=1, cnt_=cnt):
        return lib.csk_co_block_step_f32(fake, 16, 0, c_in, fake, fake, fake, fake, C.cast(cnt_, C.c_void_p), 4, gcn_res,
                                         fake, y_slots, 0, fake, fake, res_mode, 12, fake, 16, 0, c_out, n_skel, V, P, None)

    for kwargs, needle in [(dict(c_out=128), "c_out <= 64"), (dict(y_slots=9), "too shallow"), (dict(P=202), "multiple of 4"),
                           (dict(c_in=32), "identity gcn residual"), (dict(res_mode=2), "none or identity"),
                           (dict(cnt_=(C.c_int32 * 3)(2, 1, 4)), "skeleton-sparse"), (dict(V=43, P=344), "longer than 128")]:
        assert call(**kwargs) < 0
        assert needle in lib.csk_last_error().decode(), (kwargs, lib.csk_last_error().decode())


def test_no_kernel_spills():
    """Every kernel instantiation a public entry point can select is spill-free: the compiler's resource report, which
    build.sh keeps next to the library, shows 0 bytes of scratch per lane for each of them (a spilling instantiation is a
    2x performance cliff for the shapes that select it)."""
    import os
    import re
    path = os.path.join(os.path.dirname(pkg.native.LIB_PATH), "kernel_resources.txt")
    assert os.path.exists(path), "build with continual-skeletons_amd/csrc/build.sh (it writes kernel_resources.txt)"
    text = open(path).read()
    kernels = re.findall(r"Name: (\S+)\n(?:.*\n)*?ScratchSize \[bytes/lane\]: (\d+)", text)
    assert len(kernels) >= 30, len(kernels)
    spilling = [(n, int(b)) for n, b in kernels if int(b) != 0]
    assert not spilling, spilling


def test_split_weight_packing_reconstructs_the_fp32_weights():
    """fold.pack_conv_weight_split (operand image of csk_tcn_stage_bf16x3): the three bf16 pieces of every element sum back
    to the fp32 weight the exact path packs (24 significand bits), the nine taps sit in class-major order (residue classes
    modulo the stride), padding channels / rows / slots are zero."""
    from continual_skeletons_amd import fold
    g = torch.Generator().manual_seed(3)
    w = torch.randn(70, 20, 9, 1, generator=g)
    sc = torch.rand(70, generator=g).double() + 0.5
    w32 = (w.double()[:, :, :, 0] * sc[:, None, None]).float()
    for stride, order in ((1, list(range(9))), (2, [0, 2, 4, 6, 8, 1, 3, 5, 7]), (3, [0, 3, 6, 1, 4, 7, 2, 5, 8])):
        img = fold.pack_conv_weight_split(w, sc, stride)
        assert img.dtype == torch.int16 and img.numel() == 2 * 9 * 3 * 2 * 128 * 8        # [c16 = 2][9][3][2][Mpad = 128][8]
        kind = img.view(torch.bfloat16).float().view(2, 9, 3, 2, 128, 8)
        rec = kind.sum(2).permute(3, 0, 2, 4, 1).reshape(128, 32, 9)                      # [co][c][slot]
        assert torch.equal(rec[:70, :20], w32[:, :, order])                               # exact: h + m + l == the fp32 value
        assert float(rec[70:].abs().max()) == 0 and float(rec[:, 20:].abs().max()) == 0
    one = fold.pack_conv_weight_split(torch.randn(8, 5, 1, 1, generator=g), torch.ones(8, dtype=torch.float64), 2)
    assert one.numel() == 1 * 3 * 3 * 2 * 64 * 8                                          # k = 1: tap 0 + two zero slots
    assert float(one.view(torch.bfloat16).float().view(1, 3, 3, 2, 64, 8)[:, 1:].abs().max()) == 0
    three = fold.pack_conv_weight_split(torch.randn(8, 5, 3, generator=g), torch.ones(8, dtype=torch.float64))   # graph-conv subsets
    assert three.numel() == 1 * 3 * 3 * 2 * 64 * 8
    with pytest.raises(ValueError):
        fold.pack_conv_weight_split(torch.randn(8, 5, 5, 1), torch.ones(8, dtype=torch.float64))


def test_new_entry_points_validate_their_arguments_without_a_gpu():
    import ctypes as C
    lib = pkg.native.lib()
    fake = C.c_void_p(0x1000)
    assert lib.csk_conv1x1_f32(None, fake, fake, fake, 1, 4, 4, 3, 25, 300, 75, 300, 75, None) == -1
    assert b"null pointer" in lib.csk_last_error()
    assert lib.csk_conv1x1_f32(fake, fake, fake, fake, 1, 4, 4, 3, 70, 840, 210, 840, 210, None) == -1
    assert b"bad dims" in lib.csk_last_error()
    # the clip form of the attention (T > 1) needs its partial-logit scratch
    assert lib.csk_agcn_attention_f32(fake, fake, fake, None, 2, 4, 6, 18, 100, 10, 2, 0, None) == -1
    assert b"scratch" in lib.csk_last_error()
    # bf16x3 stage: the 9-tap conv only, stride <= 4, 16-byte aligned operand images
    args = dict(y=fake, w=fake, xr=None, wr=None, b=fake, o=fake)
    def split(k=9, stride=1, w=fake):
        return lib.csk_tcn_stage_bf16x3(args["y"], w, args["xr"], args["wr"], args["b"], args["o"], 1, 16, 16, 20, 25, k, stride, 4, 0, 0,
                                        0, 0, 1, None)
    assert split(k=3) == -1 and b"9 x 1" in lib.csk_last_error()
    assert split(stride=5) == -1 and b"stride" in lib.csk_last_error()
    assert split(w=C.c_void_p(0x1008)) == -1 and b"16-byte aligned" in lib.csk_last_error()


def test_set_precision_marks_blocks_and_refolds():
    m = pkg.StGcn(pkg.ntu_graph().A)
    assert all(b.precision == "f32" for b in m.layers.values())
    pkg.set_precision(m, "bf16x3")
    assert all(b.precision == "bf16x3" for b in m.layers.values())
    ops = m.layers["layer5"]._fold()                                  # stride-2 block with a conv residual
    assert ops["w_split"].dtype == torch.int16 and ops["w_res_split"] is not None and ops["w"].dtype == torch.float32
    pkg.set_precision(m, "f32")
    assert m.layers["layer5"]._fold()["w_split"] is None
    with pytest.raises(ValueError):
        pkg.set_precision(m, "fp16")
    # a block the split kernel does not cover anywhere below the module: refused BEFORE any block is switched
    m.layers["layer9"].tcn.kernel_size = 7
    with pytest.raises(NotImplementedError):
        pkg.set_precision(m, "bf16x3")
    assert all(b.precision == "f32" for b in m.layers.values())


def test_gcn_bf16x3_entry_validates_arguments_without_a_gpu():
    import ctypes as C
    lib = pkg.native.lib()
    fake = C.c_void_p(0x1000)
    cnt = (C.c_int32 * 3)(1, 1, 4)

    def call(c_out=128, cnt_=cnt, res=1, c_in=128, wres=None):
        return lib.csk_gcn_stage_bf16x3(fake, fake, fake, wres, fake, fake, fake, C.cast(cnt_, C.c_void_p), 4, 2, c_in, c_out, 10, 25,
                                        res, None)
    assert call(c_out=64) == -1 and b"multiple of 128" in lib.csk_last_error()
    assert call(cnt_=(C.c_int32 * 3)(2, 1, 4)) == -1 and b"skeleton-sparse" in lib.csk_last_error()
    assert call(c_in=64) == -1 and b"identity residual" in lib.csk_last_error()
    assert call(c_in=64, res=2) == -1 and b"conv residual without" in lib.csk_last_error()


def test_fused_attention_entry_and_plan_validate_their_arguments_without_a_gpu():
    """csk_agcn_embed_attention_f32 (shapes it is built for, alignment, scratch of the per-segment form) and the adaptive
    graph-conv fields of csk_co_layer (csk_co_plan_create) are checked before anything is launched."""
    import ctypes as C
    lib = pkg.native.lib()
    fake = C.c_void_p(0x1000)

    def call(x=fake, scratch=None, inter=16, V=18, per_frame=1, seg_stride=1000, chan_stride=100):
        return lib.csk_agcn_embed_attention_f32(x, fake, fake, fake, fake, scratch, 2, 64, inter, 4, V, per_frame, seg_stride, chan_stride, None)
    assert call(x=None) == -1 and b"null pointer" in lib.csk_last_error()
    assert call(V=20) == -1 and b"built for V in {18, 25}" in lib.csk_last_error()
    assert call(inter=24) == -1 and b"inter in {16, 32, 64}" in lib.csk_last_error()
    assert call(x=C.c_void_p(0x1004)) == -1 and b"8-byte aligned" in lib.csk_last_error()
    assert call(chan_stride=101) == -1 and b"8-byte aligned" in lib.csk_last_error()
    assert call(per_frame=0) == -1 and b"scratch" in lib.csk_last_error()
    # plan: an adaptive layer needs all four agcn_* operands and the dense ELL pattern (ell_w = V, ell_cnt = {V, V, V})
    L = (pkg.native.CoLayer * 1)()
    l = L[0]
    l.c_in, l.c_out, l.stride, l.res_kind, l.gcn_res_mode, l.ell_w, l.tcn_ksplit = 3, 64, 1, 0, 2, 18, 1
    for j in range(3):
        l.ell_cnt[j] = 18
    for f in ("gcn_w", "gcn_bias", "ell_src", "tcn_w", "tcn_bias", "y_ring", "out_ring"):
        setattr(l, f, 0x1000)
    l.agcn_inter, l.agcn_w_pairs, l.agcn_b_pairs, l.agcn_a_sum = 16, 0x1000, 0x1000, 0x1000      # agcn_adj missing
    l.y_slots, l.out_slots, l.agcn_adj_frames = 16, 8, 8
    lib.csk_co_plan_create.restype = C.c_void_p

    def create(xin0_slots=12):
        return lib.csk_co_plan_create(1, C.byref(L), fake, xin0_slots, 2, 3, 18, 2, 72, fake, fake, 400, fake, fake, 4, 1, fake, fake)
    assert not create() and b"adaptive graph conv" in lib.csk_last_error()
    l.agcn_adj = 0x1000
    l.ell_w = 6                                          # not the dense pattern
    assert not create() and b"adaptive graph conv" in lib.csk_last_error()
    l.ell_w = 18
    # ring depths (include/cskel.h: CSK_CO_Y_SLOTS / CSK_CO_IN_SLOTS) and scratch capacities are part of the contract
    assert not create(xin0_slots=4) and b"input ring needs >= 5" in lib.csk_last_error()
    # the input ring's depth fixes the largest cycle: 4 + 4 slots = cycles of <= 4 frames, for which a 12-slot y ring and a
    # 4-slot output ring (last layer: its own emissions, >= 4) suffice; the same rings are too shallow for 8-frame cycles
    l.y_slots, l.out_slots, l.agcn_adj_frames = 12, 4, 4
    small = create(xin0_slots=8)
    assert small
    lib.csk_co_plan_destroy(C.c_void_p(small))
    assert not create(xin0_slots=12) and b"rings too shallow" in lib.csk_last_error()
    l.y_slots, l.out_slots, l.agcn_adj_frames = 16, 8, 8
    l.y_slots = 15
    assert not create() and b"rings too shallow" in lib.csk_last_error()
    l.y_slots, l.out_slots = 16, 7
    assert not create() and b"rings too shallow" in lib.csk_last_error()
    l.out_slots, l.agcn_adj_frames = 8, 0
    assert not create() and b"agcn_adj_frames" in lib.csk_last_error()
    l.agcn_adj_frames, l.tcn_ksplit, l.tcn_partial, l.partial_emits = 8, 3, 0x1000, 0
    assert not create() and b"partial_emits" in lib.csk_last_error()
    l.partial_emits = 8
    plan = create()
    assert plan
    lib.csk_co_plan_destroy(C.c_void_p(plan))


def test_plan_staleness_check_is_exact_for_every_kind_of_weight_edit():
    """CoStGcn._weights_changed (host logic of the native plan): exact on the FIRST call after a replaced Parameter, a
    swapped / added sub-module, an in-place edit and ``p.data = ...`` -- checked on the snapshot the plan keeps, without a
    GPU (the plan itself is not built here)."""
    import torch
    pkg = _bootstrap.load()

    def fresh():
        net = pkg.CoStGcn(pkg.ntu_graph().A).eval()
        net.__dict__["_plan_keep"] = (None, net._weight_slots())
        assert net._weights_changed() is False and net._weights_changed() is False
        return net

    net = fresh()
    net.fc.weight = torch.nn.Parameter(net.fc.weight.detach().clone())
    assert net._weights_changed() is True
    net = fresh()
    net.layers["layer3"].tcn.bn = torch.nn.BatchNorm2d(64).eval()
    assert net._weights_changed() is True
    net = fresh()
    with torch.no_grad():
        net.fc.bias.add_(1.0)
    assert net._weights_changed() is True
    net = fresh()
    net.layers["layer10"].tcn.t_conv.weight.data = net.layers["layer10"].tcn.t_conv.weight.data.clone()
    assert net._weights_changed() is True
    net = fresh()
    net.layers["layer1"].gcn.register_buffer("extra", torch.zeros(1))
    net.layers["layer2"].add_module("extra", torch.nn.Identity())
    assert net._weights_changed() is True
    net = fresh()
    net.load_state_dict(net.state_dict())
    net._install_dirty_hooks()
    net.load_state_dict(net.state_dict())
    assert net._weights_changed() is True
    net.__dict__["_plan_keep"] = (None, net._weight_slots())          # what _refresh_plan_weights does after refolding
    assert net._weights_changed() is False


def test_round4_entry_points_validate_their_arguments_without_a_gpu():
    """csk_gcn_stage_splitk_f32, csk_co_head_step_f32, csk_input_norm_frames_f32: argument errors are reported before
    anything is launched."""
    import ctypes as C
    lib = pkg.native.lib()
    fake = C.c_void_p(0x1000)
    cnt = (C.c_int32 * 3)(1, 1, 4)

    def gcn(ksplit=4, partial=fake, ell_cnt=cnt, res=1, c_in=64, c_out=64):
        return lib.csk_gcn_stage_splitk_f32(fake, fake, fake, fake, fake, fake, ell_cnt, 4, 1, c_in, c_out, 1, 25, 1600, 25, 1600, 25, res,
                                            ksplit, partial, None)
    assert gcn(ksplit=0) == -1 and b"ksplit must be in [1, 32]" in lib.csk_last_error()
    assert gcn(ksplit=33) == -1 and b"ksplit must be in [1, 32]" in lib.csk_last_error()
    assert gcn(partial=None) == -1 and b"partial-sum buffer" in lib.csk_last_error()
    assert gcn(res=1, c_out=128) == -1 and b"identity residual" in lib.csk_last_error()
    dense = (C.c_int32 * 3)(4, 4, 4)                       # more than 1 / 1 / 4 non-zeros per column: not the sparse kernel's graph
    assert gcn(ell_cnt=dense) == -1 and b"skeleton-sparse" in lib.csk_last_error()

    def head(ring=fake, emit=1, logits=fake, count=1, head_=0, window=4):
        return lib.csk_co_head_step_f32(fake, ring, fake, fake, fake, logits, 2, 256, 50, 100, window, head_, count, emit, 60, None)
    assert head(ring=None) == -1 and b"null pointer" in lib.csk_last_error()
    assert head(head_=4) == -1 and b"bad dims" in lib.csk_last_error()
    assert head(count=5) == -1 and b"bad dims" in lib.csk_last_error()
    assert head(logits=None) == -1 and b"emitting step needs" in lib.csk_last_error()
    assert head(count=0) == -1 and b"emitting step needs" in lib.csk_last_error()

    srcs, dsts = (C.c_void_p * 2)(0x1000, 0x1000), (C.c_void_p * 2)(0x2000, 0)
    assert lib.csk_input_norm_frames_f32(srcs, dsts, 9, fake, fake, 2, 3, 25, 2, 100, None) == -1 and b"1..8 frames" in lib.csk_last_error()
    assert lib.csk_input_norm_frames_f32(srcs, dsts, 2, fake, fake, 2, 3, 25, 2, 99, None) == -1 and b"bad dims" in lib.csk_last_error()
    assert lib.csk_input_norm_frames_f32(srcs, dsts, 2, fake, fake, 2, 3, 25, 2, 100, None) == -1 and b"null frame" in lib.csk_last_error()


def test_roofline_config_prices_n_ranks_against_n_peaks():
    """bench.py passes the WHOLE JOB's work (per-rank work x world) and the step time; the accounting must price it against
    world x one MI355X's peak.  With the round-4 driver timing (72.222 ms per batch-256 step) the fraction is 0.75 at one
    rank AND at eight (weak scaling, same per-rank time) -- it used to print 6.0 at eight (round-4 review, weak item 6)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import workmodel
    fa, fe, by = workmodel.clip_totals(512)
    one = workmodel.roofline_config(fa, by, 72.222e-3, fe)
    eight = workmodel.roofline_config(8 * fa, 8 * by, 72.222e-3, 8 * fe, n_gpus=8)
    assert abs(one["frac"] - 0.7505) < 2e-3 and abs(one["frac_alg"] - 0.7775) < 2e-3
    assert eight["frac"] == one["frac"] and eight["frac_alg"] == one["frac_alg"] and 0 < eight["frac"] < 1
    assert eight["n_gpus"] == 8 and eight["peak_tflops"] == round(8 * workmodel.PEAK_F32_MFMA_TFLOPS, 1)
    assert abs(eight["achieved_tflops_per_gpu"] - one["achieved_tflops"]) < 0.02
    with pytest.raises(ValueError):
        workmodel.roofline_config(fa, by, 1.0, fe, n_gpus=0)
    # the online leg: 1024 streams, 4-frame cycle at the round-4 rate (4.12 ms per cycle) -> 0.70 at any N
    sa, se, sb = workmodel.step_totals(2048, 4)
    assert abs(workmodel.roofline_config(8 * sa, 8 * sb, 4.12e-3, 8 * se, n_gpus=8)["frac"] - workmodel.roofline_config(sa, sb, 4.12e-3, se)["frac"]) < 1e-9


def test_round5_split_k_tcn_entry_validates_its_arguments_without_a_gpu():
    """csk_tcn_stage_splitk_f32: argument errors are reported before anything is launched."""
    import ctypes as C
    lib = pkg.native.lib()
    fake = C.c_void_p(0x1000)

    def tcn(ksplit=4, partial=fake, k=9, c=64, res=1, c_res=64):
        return lib.csk_tcn_stage_splitk_f32(fake, fake, fake, None, fake, fake, 1, c, 64, 20, 25, k, 1, 4 if k == 9 else 0, res, c_res, 20, 0, 1,
                                            ksplit, partial, None)
    assert tcn(ksplit=0) == -1 and b"ksplit must be in [1, 64]" in lib.csk_last_error()
    assert tcn(ksplit=65) == -1 and b"ksplit must be in [1, 64]" in lib.csk_last_error()
    assert tcn(partial=None) == -1 and b"partial-sum buffer" in lib.csk_last_error()
    assert tcn(partial=C.c_void_p(0x1004)) == -1 and b"16-byte aligned" in lib.csk_last_error()
    assert tcn(k=1) == -1 and b"9-tap" in lib.csk_last_error()
    assert tcn(c_res=32) == -1 and b"identity residual" in lib.csk_last_error()


def test_folded_operand_cache_sees_every_kind_of_weight_edit():
    """_Folded._packed_ops re-reads a snapshot of dict slots instead of walking parameters() on every call; every way the
    weights can change must still refold on the NEXT call: in-place edit, p.data = ..., replaced Parameter, swapped
    sub-module, load_state_dict -- and an untouched module must NOT refold."""
    A = pkg.ntu_graph().A
    blk = pkg.SpatioTemporalBlock(4, 8, A, stride=2).eval()
    ops0 = blk._packed_ops("cpu")
    assert blk._packed_ops("cpu") is ops0                                   # cached
    g0 = blk.gcn._packed_ops("cpu")
    with torch.no_grad():
        blk.tcn.bn.weight.mul_(2.0)                                         # in-place: version counter
    ops1 = blk._packed_ops("cpu")
    assert ops1 is not ops0 and not torch.equal(ops1["w"], ops0["w"])
    assert blk.gcn._packed_ops("cpu") is g0                                 # the graph conv's cache watches its own tensors only
    blk.tcn.t_conv.bias.data = torch.ones_like(blk.tcn.t_conv.bias)         # p.data = ...: storage pointer
    ops2 = blk._packed_ops("cpu")
    assert ops2 is not ops1 and not torch.equal(ops2["bias"], ops1["bias"])
    blk.residual.t_conv.weight = torch.nn.Parameter(torch.zeros_like(blk.residual.t_conv.weight))    # replaced Parameter
    ops3 = blk._packed_ops("cpu")
    assert ops3 is not ops2 and float(ops3["w_res"].abs().max()) == 0.0
    blk.tcn.bn = torch.nn.BatchNorm2d(8).eval()                             # swapped sub-module
    ops4 = blk._packed_ops("cpu")
    assert ops4 is not ops3
    blk.load_state_dict({k: v.clone() + 0.25 for k, v in blk.state_dict().items()})
    assert blk._packed_ops("cpu") is not ops4
    assert blk.gcn._packed_ops("cpu") is not g0                             # load_state_dict copied into the graph conv's tensors too
    net = pkg.StGcn(A, input_shape=(3, 20, 25, 2)).eval()
    n0 = net._packed_ops("cpu")
    with torch.no_grad():
        net.fc.weight.add_(1.0)                                             # not one of the driver's folded tensors
    assert net._packed_ops("cpu") is n0
    with torch.no_grad():
        net.data_bn.running_mean.add_(1.0)
    assert net._packed_ops("cpu") is not n0


def test_co_stack_step_argument_errors_and_host_side_policies():
    """csk_co_stack_step_f32 (a run of blocks in one launch) validates the chain before any launch: block i + 1 must read what
    block i emits.  CoStGcn.set_max_cycle / SplitScratch: host-side logic that needs no GPU."""
    import ctypes as C
    import torch
    pkg = _bootstrap.load()
    lib = pkg.native.lib()

    class Args(C.Structure):
        _fields_ = [("xin", C.c_void_p), ("xin_slots", C.c_int32), ("xin_slot0", C.c_int32), ("c_in", C.c_int32),
                    ("gcn_w", C.c_void_p), ("gcn_bias", C.c_void_p), ("ell_src", C.c_void_p), ("ell_val", C.c_void_p),
                    ("ell_cnt", C.c_int32 * 3), ("ell_w", C.c_int32), ("gcn_res_mode", C.c_int32), ("y_ring", C.c_void_p),
                    ("y_slots", C.c_int32), ("y_slot0", C.c_int32), ("tcn_w", C.c_void_p), ("tcn_bias", C.c_void_p),
                    ("res_mode", C.c_int32), ("x_res_slot0", C.c_int32), ("out", C.c_void_p), ("out_slots", C.c_int32),
                    ("out_slot0", C.c_int32), ("c_out", C.c_int32)]
    blocks = (Args * 2)()
    for i, b in enumerate(blocks):
        b.xin, b.xin_slots, b.xin_slot0, b.c_in = 0x1000 * (i + 1), 12, 3, 64
        for f in ("gcn_w", "gcn_bias", "ell_src", "ell_val", "tcn_w", "tcn_bias"):
            setattr(b, f, 0x9000)
        b.ell_cnt[0], b.ell_cnt[1], b.ell_cnt[2], b.ell_w, b.gcn_res_mode = 1, 1, 4, 4, 1
        b.y_ring, b.y_slots, b.y_slot0, b.res_mode, b.x_res_slot0 = 0x5000 * (i + 1), 16, 3, 1, 11
        b.out, b.out_slots, b.out_slot0, b.c_out = 0x1000 * (i + 2), 12, 3, 64
    call = lambda n: lib.csk_co_stack_step_f32(n, C.byref(blocks), 8, 25, 200, None)
    assert call(0) == -1 and b"blocks expected" in lib.csk_last_error()
    assert call(5) == -1 and b"blocks expected" in lib.csk_last_error()
    blocks[1].xin_slot0 = 4                      # block 1 would read a slot block 0 does not write this cycle
    assert call(2) == -1 and b"does not read what block 0 emits" in lib.csk_last_error()
    blocks[1].xin_slot0, blocks[1].c_in = 3, 32
    assert call(2) == -1 and b"does not read what block 0 emits" in lib.csk_last_error()
    # set_max_cycle: range check, re-bind on change
    net = pkg.CoStGcn(pkg.ntu_graph().A).eval()
    assert net.max_cycle == 8
    net._n = 5
    net.set_max_cycle(4)
    assert net.max_cycle == 4 and net._n is None
    for bad in (0, 9, 2.5):
        with pytest.raises(ValueError):
            net.set_max_cycle(bad)
    # SplitScratch: one buffer per (device, stream), superseded buffers stay alive until release()
    from continual_skeletons_amd import blocks as blk
    sc = blk.SplitScratch()
    assert sc.nbytes() == 0
    m = pkg.StGcn(pkg.ntu_graph().A).eval()
    pkg.set_clip_latency_mode(m, 4)
    shared = {id(mod.__dict__["_split_scratch"]) for mod in m.modules() if "_split_scratch" in mod.__dict__}
    assert len(shared) == 1                      # one scratch for the whole tree
    pkg.set_clip_latency_mode(m, 0)
    assert all(mod.__dict__.get("_split_scratch") is None for mod in m.modules() if "_split_scratch" in mod.__dict__)
