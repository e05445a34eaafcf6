"""Multi-stream logit fusion and top-k accuracy: the consumer of the path's output.

Counterpart of ``scripts/multi_stream_eval.py:23-60``: predictions of 1-4 input streams (joint, bone, motion
...) stored as ``.npy`` arrays ``(N, classes)`` or ``(N, classes, steps)`` are fused with a left fold of ``np.add``
or ``np.maximum``; for step outputs the first entry is taken (``preds[:, :, 0]``, "later entries are
end-padding"); top-1/3/5 accuracies are reported.  Fusion and the per-sample rank of the target class run in
one HIP kernel (``csk_fuse_rank_f32``); file I/O stays on the host.
"""
import ctypes
import io
import pickle
from pathlib import Path
from typing import List, Sequence

import numpy as np
import torch

from . import native


class _LabelUnpickler(pickle.Unpickler):
    """A label file holds lists / tuples of str and int only: nothing in it needs a global, so none is resolved (an
    unrestricted ``pickle.load`` would run whatever callable a crafted file names; weights.py restricts its loader the
    same way)."""

    def find_class(self, module, name):
        raise pickle.UnpicklingError(f"label files hold plain lists of names and integers; refusing to load {module}.{name}")


def load_labels(label_path, trusted: bool = False) -> np.ndarray:
    """Targets of an NTU / Kinetics label file: a pickled ``(sample_names, labels)`` pair written by the reference's
    dataset tooling under Python 2 (hence latin1), as read by multi_stream_eval.py:16-20 -> int64 array (N,).
    ``trusted=True`` uses the unrestricted unpickler (label files that store numpy arrays)."""
    blob = Path(label_path).read_bytes()
    pair = pickle.loads(blob, encoding="latin1") if trusted else _LabelUnpickler(io.BytesIO(blob), encoding="latin1").load()
    if not isinstance(pair, (tuple, list)) or len(pair) != 2:
        raise ValueError(f"{label_path}: expected a pickled (sample_names, labels) pair")
    return np.asarray(pair[1], dtype=np.int64)


def load_preds(path) -> np.ndarray:
    """Stream predictions stored by the reference's test runs: a ``.npy`` array (N, classes[, steps])."""
    path = Path(path)
    if path.suffix != ".npy":
        raise ValueError(f"{path}: predictions are expected as .npy arrays")
    if not path.is_file():
        raise FileNotFoundError(path)
    arr = np.load(path, allow_pickle=False)
    if arr.ndim not in (2, 3):
        raise ValueError(f"{path}: array of shape {arr.shape}, expected (N, classes) or (N, classes, steps)")
    return arr


def _launch(preds: Sequence[torch.Tensor], method: str, targets=None, want_fused=True):
    if method not in ("add", "maximum"):
        raise ValueError("method must be 'add' or 'maximum'")
    if not 1 <= len(preds) <= 4:
        raise ValueError("1..4 prediction arrays expected")
    if len({tuple(p.shape) for p in preds}) != 1:
        # the reference asserts here (multi_stream_eval.py:35-38): same exception type
        raise AssertionError(f"prediction arrays differ in shape: {[tuple(p.shape) for p in preds]}")
    for p in preds:
        native.require_device_f32(p, "prediction array")
    p0 = preds[0]
    if p0.dim() == 3:
        n, classes, steps = p0.shape
        sample_stride, class_stride = classes * steps, steps        # preds[:, :, 0]
    elif p0.dim() == 2:
        (n, classes), sample_stride, class_stride = p0.shape, p0.shape[1], 1
    else:
        raise ValueError("predictions must be (N, classes) or (N, classes, steps)")
    ptrs = (ctypes.c_void_p * len(preds))(*[p.data_ptr() for p in preds])
    fused = torch.empty((n, classes), device=p0.device, dtype=torch.float32) if want_fused else None
    rank = None
    if targets is not None:
        targets = torch.as_tensor(targets, dtype=torch.int64, device=p0.device).contiguous()
        if targets.shape != (n,):
            raise ValueError(f"targets must have shape ({n},)")
        rank = torch.empty((n,), device=p0.device, dtype=torch.int32)
    rc = native.lib().csk_fuse_rank_f32(ptrs, len(preds), int(method == "maximum"), n, classes, sample_stride,
                                        class_stride, native.ptr(targets), native.ptr(fused), native.ptr(rank),
                                        native.stream_of(p0))
    native.check(rc, "csk_fuse_rank_f32")
    return fused, rank


def aggregate_preds(preds: List[torch.Tensor], method: str = "add") -> torch.Tensor:
    """Left fold of add / maximum over the stream predictions (multi_stream_eval.py:33-42); 3-D inputs are
    reduced to their first step."""
    return _launch(preds, method)[0]


def topk_accuracies(preds: Sequence[torch.Tensor], targets, ks=(1, 3, 5), method: str = "add"):
    """Top-k accuracies of the fused predictions.  A sample is a top-k hit when fewer than k classes score
    strictly higher than its target class."""
    _, rank = _launch(preds, method, targets=targets, want_fused=False)
    return [float((rank < k).float().mean()) for k in ks]


def multi_stream_eval(labels: str, pred1: str, pred2: str = None, pred3: str = None, pred4: str = None,
                      method: str = "add", device="cuda:0"):
    """File-level entry point with the reference's argument names (multi_stream_eval.py:45-62)."""
    paths = [p for p in (pred1, pred2, pred3, pred4) if p]
    preds = [torch.from_numpy(np.ascontiguousarray(load_preds(p), dtype=np.float32)).to(device) for p in paths]
    targets = load_labels(labels)
    accs = topk_accuracies(preds, targets, (1, 3, 5), method)
    return {f"top{k}acc": v for k, v in zip((1, 3, 5), accs)}
