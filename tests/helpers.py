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
