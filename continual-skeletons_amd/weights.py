"""Weight import path: a reference checkpoint -> the native blocks.

The reference hands ``--finetune_from_weights <file>`` to the Ride shell
(models/cost_gcn/scripts/evaluate_ntu60.py:56-57), which reads the file, takes the ``state_dict`` entry of a
Lightning ``.ckpt`` (a plain ``.pt`` is the state dict itself), passes it through the model's
``map_loaded_weights`` (models/base.py:226-227: regular ST-GCN keys -> the nested keys of the continual stack) and
loads it.  Parameter names of the native modules are the reference's, so published weights load unmodified; the
folded / packed operands of the HIP kernels are rebuilt lazily on the next forward (``blocks._Folded``).
"""
from collections import OrderedDict

import torch


def read_state_dict(path: str, trusted: bool = False) -> "OrderedDict[str, torch.Tensor]":
    """State dict stored in a ``.pt`` / ``.pth`` (plain) or ``.ckpt`` (under ``state_dict``) file, on the CPU.

    Files are read with ``weights_only=True`` (no arbitrary unpickling).  Plain state dicts and checkpoints whose
    non-tensor entries are plain containers load that way; a Lightning / Ride ``.ckpt`` usually also carries
    ``hyper_parameters`` as an ``argparse.Namespace`` / ``AttributeDict``, which the safe unpickler rejects --
    ``argparse.Namespace`` is allow-listed here, anything else needs ``trusted=True`` (full unpickling: only for
    files you trust)."""
    import argparse
    import pickle

    try:
        with torch.serialization.safe_globals([argparse.Namespace]):
            blob = torch.load(path, map_location="cpu", weights_only=not trusted)
    except pickle.UnpicklingError as e:
        raise RuntimeError(
            f"{path}: the checkpoint holds pickled objects beyond tensors, containers and argparse.Namespace (typical for "
            "Lightning 'hyper_parameters'); re-save its 'state_dict' as a plain .pt, or pass trusted=True to "
            f"load_pretrained / read_state_dict if you trust the file.  ({str(e).splitlines()[0]})") from e
    if isinstance(blob, dict) and "state_dict" in blob and isinstance(blob["state_dict"], dict):
        blob = blob["state_dict"]
    if not isinstance(blob, dict) or not all(isinstance(v, torch.Tensor) for v in blob.values()):
        raise RuntimeError(f"{path}: not a state dict (.pt) or a checkpoint with a 'state_dict' entry (.ckpt)")
    return OrderedDict(blob)


def load_pretrained(model: torch.nn.Module, path: str, strict: bool = True, trusted: bool = False):
    """Load reference weights into ``StGcn`` / ``CoStGcn`` / ``AGcn`` / ``CoAGcn`` (or any block).  Keys go through
    ``model.map_loaded_weights(path, state_dict)`` when the model defines it.  Returns what ``load_state_dict``
    returns (missing / unexpected keys when ``strict=False``).  ``trusted``: see ``read_state_dict``."""
    sd = read_state_dict(path, trusted=trusted)
    if hasattr(model, "map_loaded_weights"):
        sd = model.map_loaded_weights(path, sd) if strict else model.map_state_dict(sd, strict=False)
    return model.load_state_dict(sd, strict=strict)
