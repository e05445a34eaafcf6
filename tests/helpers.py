"""Shared helpers for the test-suite (test infrastructure; may import the oracle)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    """-> (arrays: dict[str, np.ndarray], sd: dict[str, torch.Tensor]) from tests/golden/<name>.npz"""
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    arrays = {k: d[k] for k in d.files if not k.startswith("sd/")}
    sd = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("sd/")}
    return arrays, sd


def model_fixture(name, v):
    """Rebuild the closed-form state_dict + input of a whole-model fixture (make_golden.py G6 / G8): the file stores
    only key names, shapes, seed and expected outputs."""
    from closed_form import closed_form_input, closed_form_state_dict
    from oracle import stgcn_oracle as o

    arrays, _ = load_golden(name)
    shapes = {str(k): tuple(eval(str(s))) for k, s in zip(arrays["sd_keys"], arrays["sd_shapes"])}
    scale = float(arrays["gcn_bn_scale"]) if "gcn_bn_scale" in arrays else 1.0
    gen = closed_form_state_dict(shapes, salt0=float(arrays["seed"]), gcn_bn_scale=scale)
    A = torch.from_numpy((o.ntu_graph() if v == 25 else o.kinetics_graph()).astype(np.float32))
    sd = {}
    for k in shapes:
        sd[k] = A.clone() if k.endswith(".A") else torch.from_numpy(gen[k])
    x = torch.from_numpy(closed_form_input((int(arrays["n"]), 3, 300, v, 2), salt=float(arrays["salt"])))
    return arrays, sd, x


def g6_state_dict(tag):
    """fixture g6_stgcn_<tag>: the reference's StGcn, NTU (N = 2) or Kinetics shape (N = 1)"""
    return model_fixture(f"g6_stgcn_{tag}", 25 if tag == "ntu" else 18)


def g8_state_dict():
    """fixture g8_agcn_kin: the reference's AGcn (models/a_gcn/a_gcn.py:72-145), Kinetics shape, N = 1"""
    return model_fixture("g8_agcn_kin", 18)


def max_err(a, b):
    return float((torch.as_tensor(a, dtype=torch.float64) - torch.as_tensor(b, dtype=torch.float64)).abs().max())


# ---- quantitative parity evidence -------------------------------------------------------------------------------
# Every parity assertion of the GPU suite goes through check_parity: an ABSOLUTE tolerance (north_star: 1e-4 fp32) on
# fixtures whose reference activations are O(1) (|want| <= REF_CAP, asserted), and one JSON line per check --
# {test, max_abs_err, absmax_ref, tol, ...} -- appended to gpurun_out/parity_report.jsonl (or $CSK_PARITY_REPORT), so
# the achieved margin is on record (a builder run is committed as profiles/r03_parity_report.json).
TOL = 1e-4
REF_CAP = 32.0


def _report_path():
    p = os.environ.get("CSK_PARITY_REPORT")
    if p:
        return p
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
    except OSError:
        return None
    return os.path.join(d, "parity_report.jsonl")


def check_parity(got, want, tol=TOL, ref_cap=REF_CAP, **info):
    """assert max |got - want| <= tol (absolute) and |want| <= ref_cap; record the achieved error."""
    import json

    got, want = torch.as_tensor(got), torch.as_tensor(want)
    assert tuple(got.shape) == tuple(want.shape), (tuple(got.shape), tuple(want.shape), info)
    err = max_err(got, want) if got.numel() else 0.0
    ref = float(want.double().abs().max()) if want.numel() else 0.0
    test = os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0]
    path = _report_path()
    if path:
        try:
            with open(path, "a") as f:
                f.write(json.dumps(dict(test=test, max_abs_err=err, absmax_ref=ref, tol=tol, n=int(want.numel()),
                                        **{k: (v if isinstance(v, (int, float, str, bool)) else str(v)) for k, v in info.items()})) + "\n")
        except OSError:
            pass
    assert ref <= ref_cap, f"fixture not O(1): |want| max = {ref:.3g} > {ref_cap} {info}"
    assert err <= tol, f"max |got - want| = {err:.3e} > {tol:g} (|want| max {ref:.3g}) {info}"
    return err


def unit_scale_(module, sd, want_fn, keys, target=4.0):
    """Bring a randomly initialised block's reference output to O(1) by scaling its final affine parameters (the listed
    BatchNorm weight / bias keys) -- in the module AND in the state dict handed to the oracle -- then return the new
    reference output.  The fixture changes, the tolerance does not."""
    with torch.no_grad():
        want = want_fn(sd)
        s = float(want.abs().max())
        if s > target:
            f = target / s
            own = dict(module.named_parameters())
            for k in keys:
                if k in sd:
                    sd[k] = sd[k] * f
                    own[k].mul_(f)
            want = want_fn(sd)
    return want


BLOCK_OUT_KEYS = ("tcn.bn.weight", "tcn.bn.bias", "residual.bn.weight", "residual.bn.bias")
GCN_OUT_KEYS = ("bn.weight", "bn.bias", "gcn_residual.1.weight", "gcn_residual.1.bias")


def randomise_unit_(net, seed, attn_scale=1.0):
    """Random-init weights of a whole model with O(1) activations through all ten blocks: fan-in scaled conv / linear
    weights, BN weights in [0.25, 0.75), small biases / running means (every tensor from the seeded generator).
    ``attn_scale`` scales graph_attn in [0.5, 1.5): 1 for ST-GCN (a multiplicative mask on the sparse A, base.py:262);
    about 1 / V for A-GCN, where it is ADDED to A as a dense matrix (a_gcn.py:50) and would otherwise multiply the
    activations by ~V per block."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, prm in net.named_parameters():
            if name.endswith(".A") or name == "A":
                continue
            if name.endswith("graph_attn"):
                prm.copy_((torch.rand(prm.shape, generator=g) + 0.5) * attn_scale)
            elif "bn" in name and name.endswith("weight") or name.endswith("residual.1.weight"):
                prm.copy_(torch.rand(prm.shape, generator=g) * 0.5 + 0.25)
            elif name.endswith("bias"):
                prm.copy_(torch.rand(prm.shape, generator=g) * 0.2 - 0.1)
            elif "a_conv" in name or "b_conv" in name:
                prm.copy_(torch.randn(prm.shape, generator=g) * 0.5)       # a non-uniform attention
            elif name.endswith("weight") and prm.dim() >= 2:
                prm.copy_(torch.randn(prm.shape, generator=g) * (1.0 / prm[0].numel()) ** 0.5)
        for name, buf in net.named_buffers():
            if name.endswith("running_var"):
                buf.copy_(torch.rand(buf.shape, generator=g) + 0.5)
            elif name.endswith("running_mean"):
                buf.copy_(torch.rand(buf.shape, generator=g) * 0.2 - 0.1)


class guarded_allocs:
    """Context manager: every float32 DEVICE tensor the package allocates through ``torch.empty`` / ``torch.zeros`` while it is
    active sits in the middle of a larger buffer whose ``pad`` elements on either side are NaN (``torch.empty`` results are NaN
    inside as well).  A kernel that reads outside an operand -- behind a ring, past a scratch slab, a row of weight padding that the
    operand does not have -- multiplies a NaN into its sums even where the weight is zero, so the model's output differs from an
    unguarded run; on a box where the operand happens to end a mapping the same read is a GPU memory fault (round 6)."""

    def __init__(self, pad=1 << 14):
        self.pad = pad
        self.count = 0          # guarded allocations made (a test asserts that the hook was in the path)

    def __enter__(self):
        import torch
        self._empty, self._zeros = torch.empty, torch.zeros
        pad, real_empty = self.pad, torch.empty

        def _wrap(zero):
            def alloc(*size, **kw):
                dev, dt = kw.get("device"), kw.get("dtype", torch.float32)
                on_gpu = dev is not None and str(dev).startswith("cuda")
                if not on_gpu or dt not in (None, torch.float32) or kw.get("out") is not None or kw.get("pin_memory"):
                    return (self._zeros if zero else self._empty)(*size, **kw)
                shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
                n = 1
                for s in shape:
                    n *= int(s)
                self.count += 1
                buf = real_empty((n + 2 * pad,), device=dev, dtype=torch.float32)
                buf.fill_(float("nan"))
                v = buf[pad: pad + n].view(shape)
                if zero:
                    v.zero_()
                return v
            return alloc

        torch.empty, torch.zeros = _wrap(False), _wrap(True)
        return self

    def __exit__(self, *exc):
        import torch
        torch.empty, torch.zeros = self._empty, self._zeros
        return False
