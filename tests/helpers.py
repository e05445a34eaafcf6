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


def g6_state_dict(tag):
    """Rebuild the closed-form state_dict + input of fixture g6_stgcn_<tag> (see make_golden.py G6)."""
    from closed_form import closed_form_input, closed_form_state_dict
    from oracle import stgcn_oracle as o

    arrays, _ = load_golden(f"g6_stgcn_{tag}")
    shapes = {str(k): tuple(eval(str(s))) for k, s in zip(arrays["sd_keys"], arrays["sd_shapes"])}
    gen = closed_form_state_dict(shapes, salt0=float(arrays["seed"]))
    v = 25 if tag == "ntu" else 18
    A = torch.from_numpy((o.ntu_graph() if tag == "ntu" else o.kinetics_graph()).astype(np.float32))
    sd = {}
    for k in shapes:
        sd[k] = A.clone() if k.endswith(".A") else torch.from_numpy(gen[k])
    x = torch.from_numpy(closed_form_input((int(arrays["n"]), 3, 300, v, 2), salt=float(arrays["salt"])))
    return arrays, sd, x


def max_err(a, b):
    return float((torch.as_tensor(a, dtype=torch.float64) - torch.as_tensor(b, dtype=torch.float64)).abs().max())
