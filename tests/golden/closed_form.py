"""Closed-form pseudo-random tensors shared by the fixture generator and the tests.

Both sides of a parity test regenerate inputs / large weight sets from an index formula instead of
storing megabytes or sharing an RNG stream.  Pure numpy; no reference code involved.
"""
import numpy as np


def closed_form_input(shape, salt=0.0):
    """U[0,1)-like values: frac(sin(i*12.9898 + salt*78.233) * 43758.5453), float32."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.float64)
    v = np.sin(i * 12.9898 + salt * 78.233) * 43758.5453
    return (v - np.floor(v)).astype(np.float32).reshape(shape)


def closed_form_state_dict(shapes, salt0=1000.0, gcn_bn_scale=1.0):
    """Deterministic ST-GCN-style state_dict from {key: shape} (iterated in sorted key order).

    * BatchNorm (keys ending .weight/.bias/.running_mean/.running_var next to a running_mean):
      weight U(.5,1.5) (x0.35 for ``tcn.bn`` so activations stay O(1) through 10 blocks),
      bias/mean U(-.5,.5), var U(.5,1.5); num_batches_tracked = 0.
    * graph_attn U(.5,1.5); ``A`` is NOT generated (caller keeps the graph's adjacency).
    * conv / linear weight: U(-1,1) * sqrt(3 / fan_in) * 0.9 ; bias U(-.2,.2).
    * ``gcn_bn_scale`` multiplies the ``gcn.bn`` weights (A-GCN fixture G8: the adaptive adjacency sums ~V joints with
      weights ~1, so the graph conv's BN must scale down for activations to stay O(1) through ten blocks).
    """
    keys = sorted(shapes)
    bn_prefixes = {k[: -len("running_mean")] for k in keys if k.endswith("running_mean")}
    out = {}
    for idx, k in enumerate(keys):
        shp = tuple(int(s) for s in shapes[k])
        if k.endswith(".A") or k == "A":
            continue
        u = closed_form_input(shp if shp else (1,), salt=salt0 + idx).astype(np.float64)
        if not shp:
            u = u.reshape(())
        pre = k[: k.rfind(".") + 1]
        leaf = k[k.rfind(".") + 1:]
        if leaf == "num_batches_tracked":
            out[k] = np.zeros(shp, dtype=np.int64)
        elif pre in bn_prefixes:
            if leaf == "weight":
                v = 0.5 + u
                if pre.endswith("tcn.bn."):
                    v = v * 0.35
                if pre.endswith("gcn.bn."):
                    v = v * gcn_bn_scale
            elif leaf == "running_var":
                v = 0.5 + u
            else:
                v = u - 0.5
            out[k] = v.astype(np.float32)
        elif leaf == "graph_attn":
            out[k] = (0.5 + u).astype(np.float32)
        elif leaf == "weight":
            fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else shp[0]
            out[k] = ((2 * u - 1) * np.sqrt(3.0 / fan_in) * 0.9).astype(np.float32)
        elif leaf == "bias":
            out[k] = (0.4 * u - 0.2).astype(np.float32)
        else:
            raise KeyError(f"unclassified state_dict key {k}")
    return out
