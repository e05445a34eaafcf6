#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by executing the REFERENCE's own classes.

Runs only in the build container (needs /root/reference).  Nothing here travels as code: the
outputs are data (inputs, state_dicts, expected outputs) stored as .npz.

The reference imports `ride`, `pytorch_lightning` and `continual`, none of which is installed.
They are replaced by *name-only* stubs (no arithmetic) so that the clip-path classes --
GraphConvolution, TemporalConvolution, SpatioTemporalBlock (models/base.py:230-387),
AdaptiveGraphConvolution (models/a_gcn/a_gcn.py:12-69), Graph (datasets/graph.py),
StGcn.__init__/forward (models/st_gcn/st_gcn.py:20-65) and the key map CoModelBase.map_state_dict
(models/base.py:200-224, pure string code) -- execute verbatim.

BatchNorm affine/statistics and graph_attn are RANDOMISED: with the default init
(gcn.bn.weight = 1e-6, models/base.py:256-257) the whole aggregation branch is invisible at 1e-4.

Every fixture seeds the global RNG and its own generator, so the script regenerates the committed files
bit-identically; ``--verify`` proves it (and, independently, reloads each stored state_dict into the reference class
and re-runs it on the stored input: max |diff| must be 0.0).

usage: python tests/golden/make_golden.py [--verify]
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from closed_form import closed_form_input, closed_form_state_dict  # noqa: E402
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Cfg:
        def __init__(self):
            self.names = []

        def add(self, *a, **k):
            pass

    def _empty():
        return type("_Stub", (), {})     # a fresh class each time: they are used as distinct bases

    log = types.SimpleNamespace(info=lambda *a, **k: None, warning=lambda *a, **k: None)
    ride = mod(
        "ride", getLogger=lambda *a, **k: log, Configs=_Cfg, RideModule=nn.Module,
        TopKAccuracyMetric=lambda *k: type("TopK", (), {}), SgdOneCycleOptimizer=_empty(),
        Main=lambda *a, **k: None,
    )
    ride.core = mod("ride.core", Configs=_Cfg, RideMixin=_empty(), RideClassificationDataset=_empty())
    ride.logging = mod("ride.logging", getLogger=lambda *a, **k: log)
    ride.finetune = mod("ride.finetune", Finetunable=_empty())
    ride.optimizers = mod("ride.optimizers", SgdCyclicLrOptimizer=_empty())
    mod("pytorch_lightning")
    mod("pytorch_lightning.utilities")
    mod("pytorch_lightning.utilities.parsing", AttributeDict=dict)
    mod("continual", Sequential=type("Sequential", (nn.Sequential,), {}))
    sys.path.insert(0, REF)   # the reference's `datasets` must shadow the HF package
    # datasets/__init__ pulls the Lightning data module; provide the two leaf modules only
    import importlib.util

    pkg = types.ModuleType("datasets")
    pkg.__path__ = [os.path.join(REF, "datasets")]
    sys.modules["datasets"] = pkg
    for leaf in ("graph", "ntu_rgbd", "kinetics"):
        spec = importlib.util.spec_from_file_location(f"datasets.{leaf}", os.path.join(REF, "datasets", f"{leaf}.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"datasets.{leaf}"] = m
        spec.loader.exec_module(m)
        setattr(pkg, leaf, m)
    dsmod = mod("datasets.datasets", GraphDatasets=_empty())
    pkg.datasets = dsmod


def randomise(module: nn.Module, gen: torch.Generator):
    """BN weight~U(.5,1.5), bias~U(-.5,.5), mean~U(-.5,.5), var~U(.5,1.5); graph_attn~U(.5,1.5);
    conv biases ~U(-.2,.2) (reference init sets them to 0, which would hide bias-folding bugs)."""
    def u(shape, lo, hi):
        return torch.rand(shape, generator=gen) * (hi - lo) + lo

    with torch.no_grad():
        for name, m in module.named_modules():
            if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
                m.weight.copy_(u(m.weight.shape, 0.5, 1.5))
                m.bias.copy_(u(m.bias.shape, -0.5, 0.5))
                m.running_mean.copy_(u(m.running_mean.shape, -0.5, 0.5))
                m.running_var.copy_(u(m.running_var.shape, 0.5, 1.5))
            if isinstance(m, (nn.Conv2d, nn.Linear)) and m.bias is not None:
                m.bias.copy_(u(m.bias.shape, -0.2, 0.2))
        for name, prm in module.named_parameters():
            if name.endswith("graph_attn"):
                prm.copy_(u(prm.shape, 0.5, 1.5))


def sd_np(module):
    return {"sd/" + k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


def _ref():
    """The reference's own classes (imported once, under the name-only stubs)."""
    if "models.base" not in sys.modules:
        _install_stubs()
    from datasets import kinetics, ntu_rgbd
    from models.a_gcn.a_gcn import AdaptiveGraphConvolution, AGcn
    from models.base import GraphConvolution, SpatioTemporalBlock, TemporalConvolution
    from models.st_gcn.st_gcn import StGcn
    return types.SimpleNamespace(A_ntu=ntu_rgbd.graph.A, A_kin=kinetics.graph.A, GraphConvolution=GraphConvolution,
                                 TemporalConvolution=TemporalConvolution, SpatioTemporalBlock=SpatioTemporalBlock,
                                 AdaptiveGraphConvolution=AdaptiveGraphConvolution, StGcn=StGcn, AGcn=AGcn)


def _seeded(seed):
    """Every fixture seeds BOTH the global RNG (the reference's kaiming / normal inits draw from it) and its own
    generator (randomise(), inputs), so that a fixture regenerates bit-identically."""
    torch.manual_seed(seed)
    return torch.Generator().manual_seed(seed)


G3_VARIANTS = {
    # tag: (cin, cout, stride, residual, temporal_padding)
    "nores": (4, 4, 1, False, 4),
    "ident": (4, 4, 1, True, -1),
    "convres": (2, 4, 1, True, -1),
    "strided": (2, 4, 2, True, -1),
    "nopad": (4, 4, 1, True, 0),
    "nopad_strided": (2, 4, 2, True, 0),
}
G2_VARIANTS = {"k9s1p4": (9, 1, 4), "k9s2p4": (9, 2, 4), "k1s2p0": (1, 2, 0), "k9s1p0": (9, 1, 0)}
G6_VARIANTS = {"ntu": (25, 60, 2, 106), "kin": (18, 400, 1, 107)}     # tag: (V, classes, n, seed)
G8 = dict(V=18, classes=400, n=1, seed=109, gcn_bn_scale=0.06)


def _whole_model(R, cls, A, V, classes, n, seed, gcn_bn_scale=1.0):
    """StGcn / AGcn of the reference built without its Ride shell (cls.__new__ + nn.Module.__init__ + the attributes
    the dataset mixin would provide), closed-form weights, closed-form input; -> (arrays to store, net, x)."""
    torch.manual_seed(seed)
    net = cls.__new__(cls)
    nn.Module.__init__(net)
    net.input_shape = (3, 300, V, 2)
    net.num_classes = classes
    net.graph = types.SimpleNamespace(A=A)
    cls.__init__(net, {})
    net.eval()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    gen = closed_form_state_dict(shapes, salt0=float(seed), gcn_bn_scale=gcn_bn_scale)
    full = {k: (torch.from_numpy(gen[k]) if k in gen else v) for k, v in net.state_dict().items()}
    net.load_state_dict(full)
    x = torch.from_numpy(closed_form_input((n, 3, 300, V, 2), salt=6.0 + seed))
    taps = {}
    hooks = [net.layers[f"layer{i}"].register_forward_hook(lambda mod, inp, out, i=i: taps.__setitem__(i, out))
             for i in (1, 5, 8, 10)]
    with torch.no_grad():
        logits = net(x)
    for h in hooks:
        h.remove()
    extra = {}
    for i, t in taps.items():
        extra[f"layer{i}_sub"] = t.numpy().reshape(-1)[::997].copy()
        extra[f"layer{i}_absmax"] = np.array(float(t.abs().max()))
    nparams = sum(p.numel() for p in net.parameters())
    return dict(logits=logits.numpy(), n=np.array(n), salt=np.array(6.0 + seed), seed=np.array(float(seed)),
                nparams=np.array(nparams), gcn_bn_scale=np.array(gcn_bn_scale), sd_keys=np.array(list(shapes.keys())),
                sd_shapes=np.array([str(list(v)) for v in shapes.values()]), **extra)


def generate():
    """name -> dict of arrays, produced by the reference's classes.  Deterministic (see _seeded)."""
    R = _ref()
    torch.set_num_threads(4)
    out = {}
    # G0 -- adjacency, exact
    out["g0_graphs"] = dict(ntu=R.A_ntu, kinetics=R.A_kin)
    # G1 -- GraphConvolution 4->4 (identity gcn_residual) and 3->8 (conv gcn_residual), (2,C,6,25)
    for j, (tag, (ci, co)) in enumerate({"eq": (4, 4), "neq": (3, 8)}.items()):
        g = _seeded(1010 + j)
        m = R.GraphConvolution(ci, co, R.A_ntu).eval()
        randomise(m, g)
        x = torch.rand((2, ci, 6, 25), generator=g)
        with torch.no_grad():
            y = m(x)
        out[f"g1_gcn_{tag}"] = dict(x=x.numpy(), y=y.numpy(), **sd_np(m))
    # G2 -- TemporalConvolution variants, (2,4,20,25) (mirrors tests/test_cost_gcn.py:37-68)
    for j, (tag, (k, s, p)) in enumerate(G2_VARIANTS.items()):
        g = _seeded(1020 + j)
        m = R.TemporalConvolution(4, 4, k, s, p).eval()
        randomise(m, g)
        x = torch.rand((2, 4, 20, 25), generator=g)
        with torch.no_grad():
            y = m(x)
        out[f"g2_tcn_{tag}"] = dict(x=x.numpy(), y=y.numpy(), meta=np.array([k, s, p]), **sd_np(m))
    # G3 -- SpatioTemporalBlock variants, T=20, B=2, V=25 (mirrors tests/test_cost_gcn.py:71-271,
    #       tests/test_st_gcn_mod.py:11-54); each also serves as the step oracle via the index map
    for j, (tag, (ci, co, s, res, tp)) in enumerate(G3_VARIANTS.items()):
        g = _seeded(1030 + j)
        m = R.SpatioTemporalBlock(ci, co, R.A_ntu, s, res, temporal_padding=tp).eval()
        randomise(m, g)
        x = torch.rand((2, ci, 20, 25), generator=g)
        with torch.no_grad():
            y = m(x)
        out[f"g3_block_{tag}"] = dict(x=x.numpy(), y=y.numpy(), meta=np.array([ci, co, s, int(res), tp]), **sd_np(m))
    # G4 -- 3-block stack 3->3 (no res) ->3 (identity) ->4 (stride 2), T=40 (tests/test_cost_gcn.py:274-326)
    g = _seeded(104)
    stack = nn.Sequential(
        R.SpatioTemporalBlock(3, 3, R.A_ntu, residual=False),
        R.SpatioTemporalBlock(3, 3, R.A_ntu),
        R.SpatioTemporalBlock(3, 4, R.A_ntu, stride=2),
    ).eval()
    randomise(stack, g)
    x = torch.rand((2, 3, 40, 25), generator=g)
    with torch.no_grad():
        y = stack(x)
    out["g4_stack"] = dict(x=x.numpy(), y=y.numpy(), **sd_np(stack))
    # G5 -- BASELINE config 1: SpatioTemporalBlock(3,64,A_ntu,residual=False) on (2,3,300,25)
    g = _seeded(105)
    m = R.SpatioTemporalBlock(3, 64, R.A_ntu, residual=False).eval()
    randomise(m, g)
    x = torch.from_numpy(closed_form_input((2, 3, 300, 25), salt=5.0))
    with torch.no_grad():
        y = m(x)
    out["g5_config1_block"] = dict(y_sub7=y.numpy().reshape(-1)[::7].copy(), y_chan_sum=y.sum(dim=(0, 2, 3)).numpy(),
                                   y_shape=np.array(y.shape), **sd_np(m))
    # G6 -- full StGcn, NTU N=2 and Kinetics-shape N=1; closed-form weights and input; logits in full,
    #       layers 1/5/8/10 subsampled
    for tag, (V, classes, n, seed) in G6_VARIANTS.items():
        out[f"g6_stgcn_{tag}"] = _whole_model(R, R.StGcn, R.A_ntu if V == 25 else R.A_kin, V, classes, n, seed)
    # G7 -- AdaptiveGraphConvolution 3->8 and 8->8, V=18, T in {1,6} (T=1 is the CoAGCN step oracle)
    for j, (tag, (ci, co)) in enumerate({"neq": (3, 8), "eq": (8, 8)}.items()):
        g = _seeded(1080 + j)
        m = R.AdaptiveGraphConvolution(ci, co, R.A_kin).eval()
        randomise(m, g)
        o_ = {}
        for t in (1, 6):
            x = torch.rand((2, ci, t, 18), generator=g)
            with torch.no_grad():
                o_[f"x_t{t}"] = x.numpy()
                o_[f"y_t{t}"] = m(x).numpy()
        out[f"g7_agcn_{tag}"] = dict(**o_, **sd_np(m))
    # G8 -- full AGcn (models/a_gcn/a_gcn.py:72-145), Kinetics shape, N=1: closed-form weights including non-trivial
    #       a_conv / b_conv (the per-sample attention), logits + layer 1/5/8/10 taps
    out["g8_agcn_kin"] = _whole_model(R, R.AGcn, R.A_kin, G8["V"], G8["classes"], G8["n"], G8["seed"], G8["gcn_bn_scale"])
    # G9 -- CoModelBase.map_state_dict (models/base.py:200-224), the reference's regular -> continual key map: pure
    #       string code, executed UNBOUND on a stub nn.Module whose state_dict keys are the continual key layout.
    #       What is EXECUTED is the mapping function; the continual key LAYOUT (co_keys) is DERIVED by _co_key below from
    #       the container structure of base.py:412-446 as the reference's tests spell it out -- continual-inference is not
    #       installable here, so it is not the state_dict() of a real CoStGcn (buffers co.Delay / co.Conv2d may register
    #       are not covered).  Regenerate co_keys from the real class if the library ever becomes available.
    out["g9_key_map"] = _key_map(R)
    # G10 -- aggregate_preds (scripts/multi_stream_eval.py:33-42), the reference's own multi-stream logit fusion, executed
    #        under the same name-only stubs: np.add / np.maximum over 1-4 streams, 2-D (N, classes) and 3-D (N, classes,
    #        steps) predictions (the 3-D rule ``preds[:, :, 0]`` is multi_stream_eval.py:56-57, applied by the caller)
    out["g10_fusion"] = _fusion(R)
    return out


def _fusion(R):
    import importlib.util
    sys.modules["ride"].metrics = types.ModuleType("ride.metrics")
    sys.modules["ride.metrics"] = sys.modules["ride"].metrics
    sys.modules["ride.metrics"].topk_accuracies = lambda *a, **k: None          # a name only: never called here
    spec = importlib.util.spec_from_file_location("ref_multi_stream_eval", os.path.join(REF, "scripts", "multi_stream_eval.py"))
    mse = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mse)
    rng = np.random.default_rng(1010)
    out = {}
    for dims, shape in (("2d", (37, 60)), ("3d", (11, 60, 5))):
        preds = [rng.standard_normal(shape).astype(np.float32) for _ in range(4)]
        preds[1][3, 7] = preds[0][3, 7]                                 # a tie between streams
        preds[2][5] = np.float32(-0.0) if dims == "2d" else preds[2][5]  # signed zeros under maximum / add
        for i, pr in enumerate(preds):
            out[f"{dims}/pred{i}"] = pr
        for n in (1, 2, 3, 4):
            out[f"{dims}/add{n}"] = mse.aggregate_preds(preds[:n], np.add)
            out[f"{dims}/maximum{n}"] = mse.aggregate_preds(preds[:n], np.maximum)
    return out


def _co_key(k: str) -> str:
    """Continual-layout name of a regular ST-GCN key, from the container structure of CoSpatioTemporalBlock
    (models/base.py:412-446) as the reference's tests spell it out (tests/test_cost_gcn.py:97-98,145,193-198,299-311):
    residual=False -> Sequential(gcn, tcn, relu): unchanged; identity residual -> Sequential(Residual(Sequential(gcn,
    tcn)), relu): ``0.1.gcn`` / ``0.1.tcn``; conv residual -> Sequential(BroadcastReduce(Sequential(residual, align),
    Sequential(gcn, tcn)), relu): ``0.0.residual`` and ``0.1.gcn`` / ``0.1.tcn``."""
    parts = k.split(".")
    if parts[0] != "layers" or parts[1] == "layer1":                       # layer1: residual=False (st_gcn.py:30)
        return k
    head, rest = ".".join(parts[:2]), parts[2:]
    return ".".join([head, "0", "0" if rest[0] == "residual" else "1"] + rest)


def _key_map(R):
    from models.base import CoModelBase
    reg = R.StGcn.__new__(R.StGcn)
    nn.Module.__init__(reg)
    reg.input_shape, reg.num_classes, reg.graph = (3, 300, 25, 2), 60, types.SimpleNamespace(A=R.A_ntu)
    torch.manual_seed(9)
    R.StGcn.__init__(reg, {})
    regular = list(reg.state_dict().keys())
    co_keys = [_co_key(k) for k in regular]
    stub = nn.Module()                                     # a module tree whose state_dict() has exactly the continual keys
    for k in co_keys:
        m, parts = stub, k.split(".")
        for prt in parts[:-1]:
            if prt not in m._modules:
                m.add_module(prt, nn.Module())
            m = m._modules[prt]
        m.register_buffer(parts[-1], torch.zeros(1))
    assert list(nn.Module.state_dict(stub).keys()) == co_keys
    sd = {k: i for i, k in enumerate(regular)}             # values: positions, so that the fixture also pins the pairing
    strict = CoModelBase.map_state_dict(stub, dict(sd), strict=True)
    loose = CoModelBase.map_state_dict(stub, dict(sd, **{"not.a.key": -1}), strict=False)
    same = CoModelBase.map_state_dict(stub, {k: i for i, k in enumerate(co_keys)}, strict=True)   # already continual: returned as is
    return dict(regular_keys=np.array(regular), co_keys=np.array(co_keys),
                mapped_strict_keys=np.array(list(strict.keys())), mapped_strict_pos=np.array(list(strict.values())),
                mapped_loose_keys=np.array(list(loose.keys())), mapped_loose_pos=np.array(list(loose.values())),
                mapped_same_keys=np.array(list(same.keys())))


def verify():
    """Re-run the reference on every COMMITTED fixture: (a) regenerate all fixtures and demand bit-identical arrays;
    (b) independently, reload each stored state_dict into the reference class, feed the stored input and demand
    max |diff| == 0.0 against the stored output.  Exit status 1 on any difference."""
    R = _ref()
    torch.set_num_threads(4)
    bad = []

    def load(name):
        d = np.load(os.path.join(OUT, name + ".npz"))
        return {k: d[k] for k in d.files}, {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("sd/")}

    def check(name, got, want):
        diff = float(np.abs(np.asarray(got, dtype=np.float64) - np.asarray(want, dtype=np.float64)).max())
        print(f"  {name:28s} max |reference(stored sd, stored x) - stored y| = {diff}")
        if diff != 0.0:
            bad.append(name)

    with torch.no_grad():
        for tag, (ci, co) in {"eq": (4, 4), "neq": (3, 8)}.items():
            a, sd = load(f"g1_gcn_{tag}")
            m = R.GraphConvolution(ci, co, R.A_ntu).eval()
            m.load_state_dict(sd, strict=True)
            check(f"g1_gcn_{tag}", m(torch.from_numpy(a["x"])).numpy(), a["y"])
        for tag, (k, s, p) in G2_VARIANTS.items():
            a, sd = load(f"g2_tcn_{tag}")
            m = R.TemporalConvolution(4, 4, k, s, p).eval()
            m.load_state_dict(sd, strict=True)
            check(f"g2_tcn_{tag}", m(torch.from_numpy(a["x"])).numpy(), a["y"])
        for tag, (ci, co, s, res, tp) in G3_VARIANTS.items():
            a, sd = load(f"g3_block_{tag}")
            m = R.SpatioTemporalBlock(ci, co, R.A_ntu, s, res, temporal_padding=tp).eval()
            m.load_state_dict(sd, strict=True)
            check(f"g3_block_{tag}", m(torch.from_numpy(a["x"])).numpy(), a["y"])
        a, sd = load("g4_stack")
        stack = nn.Sequential(R.SpatioTemporalBlock(3, 3, R.A_ntu, residual=False), R.SpatioTemporalBlock(3, 3, R.A_ntu),
                              R.SpatioTemporalBlock(3, 4, R.A_ntu, stride=2)).eval()
        stack.load_state_dict(sd, strict=True)
        check("g4_stack", stack(torch.from_numpy(a["x"])).numpy(), a["y"])
        a, sd = load("g5_config1_block")
        m = R.SpatioTemporalBlock(3, 64, R.A_ntu, residual=False).eval()
        m.load_state_dict(sd, strict=True)
        y = m(torch.from_numpy(closed_form_input((2, 3, 300, 25), salt=5.0)))
        check("g5_config1_block", y.numpy().reshape(-1)[::7], a["y_sub7"])
        for tag, (ci, co) in {"neq": (3, 8), "eq": (8, 8)}.items():
            a, sd = load(f"g7_agcn_{tag}")
            m = R.AdaptiveGraphConvolution(ci, co, R.A_kin).eval()
            m.load_state_dict(sd, strict=True)
            for t in (1, 6):
                check(f"g7_agcn_{tag} T={t}", m(torch.from_numpy(a[f"x_t{t}"])).numpy(), a[f"y_t{t}"])
    # (a) everything, including G0 / G6 / G8 (whose weights are closed-form, not stored): regenerate and compare
    fresh = generate()
    for name, arrays in fresh.items():
        d = np.load(os.path.join(OUT, name + ".npz"))
        same = set(d.files) == set(arrays) and all(
            np.array_equal(np.asarray(arrays[k]), d[k]) for k in d.files)
        print(f"  {name:28s} regenerates bit-identically: {same}")
        if not same:
            bad.append(name + " (regeneration)")
    if bad:
        print("MISMATCH:", bad)
        raise SystemExit(1)
    print("all committed fixtures verified against the reference: 0.0")


def main():
    if "--verify" in sys.argv[1:]:
        return verify()
    for name, arrays in generate().items():
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT) if f.endswith(".npz"))
    print(f"wrote fixtures to {OUT}: {tot / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
