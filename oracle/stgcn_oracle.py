"""CPU oracle for the ST-GCN / CoST-GCN forward path.  TEST INFRASTRUCTURE ONLY.

This file restates, with stock PyTorch CPU ops in the same op order, the arithmetic of the
reference's block library.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; the product package (``continual-skeletons_amd/``) never does.

Parity status
-------------
* clip path (graph conv, temporal conv, block, 10-block ST-GCN): **pinned** against outputs of the
  reference's own classes executed in the build container (``tests/golden/make_golden.py`` imports
  ``/root/reference`` under name-only stubs and writes ``tests/golden/*.npz``);
  ``tests/test_oracle_golden.py`` replays them.
* continual (``forward_step``) path: the arithmetic lives in the un-vendored third-party package
  ``continual-inference>=0.16.0`` (reference ``requirements.txt:3``), absent from ``/root/reference``
  and not installable here.  Its published behaviour is restated below and anchored on the
  reference's own tests (``tests/test_cost_gcn.py:37-326``, ``tests/test_st_gcn_mod.py:11-54``),
  which all have the form *continual output == clip output at a shifted frame index*; the clip side
  of that identity is pinned as above.
* model-level continual head defaults (``pool_size``/``pool_padding`` = -1, ``models/base.py:86-97``)
  depend on library properties no exact reference test pins: **parity unpinned** for that default;
  ``co_stgcn_pool_defaults`` documents the derived values (75 / 19 for NTU-60).

All functions are functional: ``sd`` is a reference-layout ``state_dict`` (dict of tensors),
``p`` the key prefix of the module.
"""
from __future__ import annotations

import math
from collections import deque
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
BN_EPS = 1e-5  # torch.nn.BatchNorm{1,2}d default, used unchanged by the reference

# --------------------------------------------------------------------------------------------
# Skeleton graphs (reference: datasets/graph.py:9-44, datasets/ntu_rgbd.py:3-35, datasets/kinetics.py:24-46)
# --------------------------------------------------------------------------------------------
# NTU RGB+D bone list, 1-based (child, parent) pairs as published with the dataset.
_NTU_BONES_1B = (
    (1, 2), (2, 21), (3, 21), (4, 3), (5, 21), (6, 5), (7, 6), (8, 7), (9, 21), (10, 9), (11, 10),
    (12, 11), (13, 1), (14, 13), (15, 14), (16, 15), (17, 1), (18, 17), (19, 18), (20, 19),
    (22, 23), (23, 8), (24, 25), (25, 12),
)
NTU_INWARD = tuple((a - 1, b - 1) for a, b in _NTU_BONES_1B)
NTU_V = 25
# OpenPose-18 bone list, 0-based (origin, neighbour).
KINETICS_INWARD = (
    (4, 3), (3, 2), (7, 6), (6, 5), (13, 12), (12, 11), (10, 9), (9, 8), (11, 5), (8, 2), (5, 1),
    (2, 1), (0, 1), (15, 0), (14, 0), (17, 15), (16, 14),
)
KINETICS_V = 18


def spatial_graph(inward: Sequence[Tuple[int, int]], num_node: int) -> np.ndarray:
    """A = stack(I, In, Out), float64 (datasets/graph.py:27-32).

    ``edge2mat`` sets M[j, i] = 1 for a link (i, j) (graph.py:9-13); ``normalize_digraph`` divides
    every column by its sum when that is > 0 (graph.py:16-24); outward = reversed inward (graph.py:40).
    """
    def mat(links):
        m = np.zeros((num_node, num_node))
        for i, j in links:
            m[j, i] = 1.0
        return m

    def colnorm(m):
        s = m.sum(0)
        d = np.zeros_like(s)
        d[s > 0] = 1.0 / s[s > 0]
        return m @ np.diag(d)

    eye = mat([(i, i) for i in range(num_node)])
    inw = colnorm(mat(inward))
    outw = colnorm(mat([(j, i) for i, j in inward]))
    return np.stack((eye, inw, outw))


def ntu_graph() -> np.ndarray:
    return spatial_graph(NTU_INWARD, NTU_V)


def kinetics_graph() -> np.ndarray:
    return spatial_graph(KINETICS_INWARD, KINETICS_V)


# --------------------------------------------------------------------------------------------
# Block library, clip mode (reference: models/base.py:230-387)
# --------------------------------------------------------------------------------------------
def _bn(x: Tensor, sd: Dict[str, Tensor], p: str) -> Tensor:
    """Eval-mode batch norm with the module's running statistics."""
    return F.batch_norm(
        x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"],
        False, 0.0, BN_EPS,
    )


def graph_conv(x: Tensor, sd: Dict[str, Tensor], p: str = "") -> Tensor:
    """GraphConvolution.forward (models/base.py:260-270).  x: (N, C, T, V)."""
    n, c, t, v = x.shape
    a_eff = sd[p + "A"] * sd[p + "graph_attn"]                     # base.py:262
    acc = None
    for i in range(3):                                             # base.py:264
        xa = torch.matmul(x.reshape(n, c * t, v), a_eff[i]).view(n, c, t, v)   # base.py:265-266
        z = F.conv2d(xa, sd[f"{p}g_conv.{i}.weight"], sd[f"{p}g_conv.{i}.bias"])
        acc = z if acc is None else z + acc                       # base.py:267
    acc = _bn(acc, sd, p + "bn.")                                  # base.py:268
    if (p + "gcn_residual.0.weight") in sd:                        # base.py:246-254
        r = F.conv2d(x, sd[p + "gcn_residual.0.weight"], sd[p + "gcn_residual.0.bias"])
        r = _bn(r, sd, p + "gcn_residual.1.")
    else:
        r = x
    return F.relu(acc + r)                                         # base.py:269-270


def temporal_conv(x: Tensor, sd: Dict[str, Tensor], p: str = "", stride: int = 1,
                  padding: int = 4) -> Tensor:
    """TemporalConvolution.forward (models/base.py:302-304): BN(conv (k,1), stride (s,1), pad (p,0))."""
    z = F.conv2d(x, sd[p + "t_conv.weight"], sd[p + "t_conv.bias"], stride=(stride, 1),
                 padding=(padding, 0))
    return _bn(z, sd, p + "bn.")


def block_kind(sd: Dict[str, Tensor], p: str, residual: bool) -> str:
    """'none' | 'identity' | 'conv' -- the three residual forms of base.py:367-374."""
    if not residual:
        return "none"
    return "conv" if (p + "residual.t_conv.weight") in sd else "identity"


def st_block(x: Tensor, sd: Dict[str, Tensor], p: str = "", stride: int = 1, residual: bool = True,
             temporal_padding: int = -1, gcn=graph_conv) -> Tensor:
    """SpatioTemporalBlock.forward (models/base.py:376-387)."""
    k = sd[p + "tcn.t_conv.weight"].shape[2]
    equal = (k - 1) // 2
    if temporal_padding < 0:                                       # base.py:352-354
        temporal_padding, shrink = equal, 0
    else:                                                          # base.py:355-357
        assert temporal_padding <= equal
        shrink = equal - temporal_padding
    z = temporal_conv(gcn(x, sd, p + "gcn."), sd, p + "tcn.", stride, temporal_padding)
    kind = block_kind(sd, p, residual)
    xr = x[:, :, shrink:x.shape[2] - shrink] if shrink else x      # base.py:379-385
    if kind == "none":
        r = 0
    elif kind == "identity":
        r = xr
    else:
        r = temporal_conv(xr, sd, p + "residual.", stride, 0)      # base.py:372-374 (k=1, pad 0)
    return F.relu(z + r)                                           # base.py:387


# (in, out, stride, residual) -- models/st_gcn/st_gcn.py:30-39 == models/cost_gcn/cost_gcn.py:31-40
def layer_table(c_in: int = 3):
    return [
        (c_in, 64, 1, False), (64, 64, 1, True), (64, 64, 1, True), (64, 64, 1, True),
        (64, 128, 2, True), (128, 128, 1, True), (128, 128, 1, True),
        (128, 256, 2, True), (256, 256, 1, True), (256, 256, 1, True),
    ]


def stgcn_pre(x: Tensor, sd: Dict[str, Tensor]) -> Tensor:
    """Input permute + data_bn + reshape to (N*M, C, T, V) (models/st_gcn/st_gcn.py:49-57)."""
    n, c, t, v, m = x.shape
    h = x.permute(0, 4, 3, 1, 2).contiguous().view(n, m * v * c, t)
    h = _bn(h, sd, "data_bn.")
    return h.view(n, m, v, c, t).permute(0, 1, 3, 4, 2).contiguous().view(n * m, c, t, v)


def stgcn_head(h: Tensor, sd: Dict[str, Tensor], n: int, m: int) -> Tensor:
    """mean over (T*V), mean over M, fc (models/st_gcn/st_gcn.py:60-64)."""
    c = h.shape[1]
    h = h.view(n, m, c, -1).mean(3).mean(1)
    return F.linear(h, sd["fc.weight"], sd["fc.bias"])


def stgcn_forward(x: Tensor, sd: Dict[str, Tensor], gcn=graph_conv, taps=None) -> Tensor:
    """StGcn.forward (models/st_gcn/st_gcn.py:48-65).  x: (N, C, T, V, M) -> (N, classes).

    ``taps`` (optional dict) receives the activations after each layer, keyed ``layerK``.
    """
    n, c, t, v, m = x.shape
    h = stgcn_pre(x, sd)
    for i, (_, _, stride, res) in enumerate(layer_table(c)):
        h = st_block(h, sd, f"layers.layer{i + 1}.", stride, res, gcn=gcn)
        if taps is not None:
            taps[f"layer{i + 1}"] = h
    return stgcn_head(h, sd, n, m)


def co_stgcn_steps_pad_end(x: Tensor, sd: Dict[str, Tensor], pool_size: int, pool_padding: int, gcn=graph_conv) -> Tensor:
    """What ``CoModelBase.forward_steps(x, pad_end=True)`` (models/base.py:187-190) yields from a clean state, written
    in clip form: with ``pad_end`` every continual conv is flushed with its end padding, so each block emits exactly
    the 'same'-padded clip block's frames (the identity of tests/test_cost_gcn.py:37-68 applied block by block);
    ``spatial_pool`` per frame (base.py:84); ``AvgPool1d(pool_size, stride 1, padding pool_padding)`` with the zero
    padding counted in the divisor (base.py:97: zero-initialised window in front, flushed zeros behind); ``Linear``
    per step (base.py:99).  x: (N, C, T, V, M) -> (N, classes, n_predictions)."""
    n, c, t, v, m = x.shape
    h = stgcn_pre(x, sd)
    for i, (_, _, stride, res) in enumerate(layer_table(c)):
        h = st_block(h, sd, f"layers.layer{i + 1}.", stride, res, gcn=gcn)
    f = h.view(n, m, h.shape[1], h.shape[2], v).mean(4).mean(1)                       # (N, 256, T')
    pooled = F.avg_pool1d(f, pool_size, 1, padding=pool_padding, count_include_pad=True)
    return torch.einsum("nct,kc->nkt", pooled, sd["fc.weight"]) + sd["fc.bias"][None, :, None]


# --------------------------------------------------------------------------------------------
# A-GCN adaptive graph convolution (reference: models/a_gcn/a_gcn.py:12-69)
# --------------------------------------------------------------------------------------------
def adaptive_graph_conv(x: Tensor, sd: Dict[str, Tensor], p: str = "") -> Tensor:
    """AdaptiveGraphConvolution.forward (models/a_gcn/a_gcn.py:48-69)."""
    n, c, t, v = x.shape
    a_sum = sd[p + "A"] + sd[p + "graph_attn"]                                     # a_gcn.py:50
    acc = None
    for i in range(3):
        a1 = F.conv2d(x, sd[f"{p}a_conv.{i}.weight"], sd[f"{p}a_conv.{i}.bias"])
        inter = a1.shape[1]
        a1 = a1.permute(0, 3, 1, 2).contiguous().view(n, v, inter * t)             # a_gcn.py:53-58
        a2 = F.conv2d(x, sd[f"{p}b_conv.{i}.weight"], sd[f"{p}b_conv.{i}.bias"]).view(n, inter * t, v)
        attn = torch.softmax(torch.matmul(a1, a2) / a1.size(-1), dim=-2) + a_sum[i]  # a_gcn.py:62-63
        xa = torch.matmul(x.reshape(n, c * t, v), attn).view(n, c, t, v)            # a_gcn.py:64-65
        z = F.conv2d(xa, sd[f"{p}g_conv.{i}.weight"], sd[f"{p}g_conv.{i}.bias"])
        acc = z if acc is None else z + acc
    acc = _bn(acc, sd, p + "bn.")
    if (p + "gcn_residual.0.weight") in sd:
        r = F.conv2d(x, sd[p + "gcn_residual.0.weight"], sd[p + "gcn_residual.0.bias"])
        r = _bn(r, sd, p + "gcn_residual.1.")
    else:
        r = x
    return F.relu(acc + r)


# --------------------------------------------------------------------------------------------
# Continual (frame-by-frame) protocol -- restated, see module docstring for its anchoring.
# --------------------------------------------------------------------------------------------
class CoBlockOracle:
    """One CoSpatioTemporalBlock (models/base.py:390-446) as an explicit state machine.

    Per step ``s`` (0-based count of frames this block has received since ``clean_state``):
      * ``y_s = graph_conv(x_s)`` -- ``co.forward_stepping`` applies the module per frame, stateless
        (base.py:273-276).
      * the temporal conv keeps the last k-1 post-GCN frames, zero-initialised (acts as the clip's
        left zero padding); its output at step s is the conv over frames s-(k-1)..s, i.e. the clip
        output at t = s - (k-1-p) (tests/test_cost_gcn.py:118-126).  Nothing is emitted while
        s < delay = k-1-p; with temporal stride S only steps with (s - delay) % S == 0 emit
        (clip index 0 is the first emission; tests/test_cost_gcn.py:266-267).
      * residual: identity -> x delayed by ``delay`` steps (co.Residual, base.py:415-422;
        tests/test_cost_gcn.py:131-176); conv -> BN(conv1x1(x)) evaluated on the steps the k=1
        stride-S conv emits (s % S == 0) and delayed by delay//S emissions (base.py:424-446), which
        pairs emission s with x frame s - delay.  With ``padding=0`` (the "*" variants) delay = k-1,
        the block output at step s is the un-padded clip output t = s-(k-1) and the residual is the
        centred frame s - (k-1)/2 (base.py:426-430; tests/test_st_gcn_mod.py:11-54).
    ``forward_steps(x, pad_end=True)`` additionally pushes ``p`` zero post-GCN frames so that the
    whole clip output is produced (tests/test_cost_gcn.py:66-68,223-224); ``pad_end=False`` yields
    the clip output minus its last delay//S frames (tests/test_cost_gcn.py:219-220,266-267).
    """

    def __init__(self, sd: Dict[str, Tensor], p: str, stride: int = 1, residual: bool = True,
                 padding: int = 4, gcn=graph_conv):
        self.sd, self.p, self.stride, self.gcn = sd, p, stride, gcn
        self.kind = block_kind(sd, p, residual)
        self.k = sd[p + "tcn.t_conv.weight"].shape[2]
        self.padding = padding
        self.delay = self.k - 1 - padding              # steps before the first emission
        self.clean_state()

    # emission s <-> clip index (s - delay) / stride ; residual frame index = s - delay + shrink
    def clean_state(self):
        self.s = 0
        self.ring: deque = deque(maxlen=self.k)         # post-GCN frames (None == zero frame)
        self.xhist: deque = deque(maxlen=self.k)        # raw input frames for the residual

    def _tconv_at(self, frames: List[Optional[Tensor]]) -> Tensor:
        """BN(conv) over exactly k frames (None = zero frame)."""
        ref = next(f for f in frames if f is not None)
        stack = torch.stack([f if f is not None else torch.zeros_like(ref) for f in frames], dim=2)
        z = F.conv2d(stack, self.sd[self.p + "tcn.t_conv.weight"], self.sd[self.p + "tcn.t_conv.bias"])
        return _bn(z, self.sd, self.p + "tcn.bn.")[:, :, 0]

    def _residual(self, xf: Tensor):
        if self.kind == "identity":
            return xf
        r = temporal_conv(xf.unsqueeze(2), self.sd, self.p + "residual.", 1, 0)
        return r[:, :, 0]

    def _emit(self, s: int) -> Optional[Tensor]:
        """Output belonging to step index s (may lie in the flushed tail), or None."""
        if s < self.delay or (s - self.delay) % self.stride != 0:
            return None
        frames = list(self.ring)
        frames = [None] * (self.k - len(frames)) + frames
        z = self._tconv_at(frames)
        if self.kind != "none":
            shrink = (self.k - 1) // 2 - self.padding
            lag = self.delay - shrink                   # x frame index = s - lag
            xs = list(self.xhist)
            xf = xs[len(xs) - 1 - lag]                  # xhist[-1] is frame s (or a None pad)
            z = z + self._residual(xf)
        return F.relu(z)

    def forward_step(self, x_t: Tensor) -> Optional[Tensor]:
        """x_t: (N, C_in, V) -> (N, C_out, V) or None when this step emits nothing."""
        y = self.gcn(x_t.unsqueeze(2), self.sd, self.p + "gcn.")[:, :, 0]
        self.ring.append(y)
        self.xhist.append(x_t)
        out = self._emit(self.s)
        self.s += 1
        return out

    def forward_steps(self, x: Tensor, pad_end: bool = False) -> Tensor:
        outs = [o for o in (self.forward_step(x[:, :, t]) for t in range(x.shape[2])) if o is not None]
        if pad_end:
            for _ in range(self.padding):               # flush with zero post-GCN frames
                self.ring.append(None)
                self.xhist.append(None)
                o = self._emit(self.s)
                self.s += 1
                if o is not None:
                    outs.append(o)
        return torch.stack(outs, dim=2)


def co_stgcn_geometry(c_in: int = 3):
    """Receptive field / padding / stride of the 10-block continual stack (properties the reference
    reads from co.Sequential at models/base.py:86-97), derived from the layer table:
    R accumulates (k-1)*cumulative_stride, P accumulates p*cumulative_stride."""
    r, p, s = 1, 0, 1
    for (_, _, st, _) in layer_table(c_in):
        r += 8 * s
        p += 4 * s
        s *= st
    return r, p, s          # NTU/Kinetics: (153, 76, 4)


def co_stgcn_pool_defaults(t: int = 300, c_in: int = 3):
    """pool_size / pool_padding defaults of CoModelBase.on_init_end (models/base.py:86-97)."""
    r, p, s = co_stgcn_geometry(c_in)
    size = math.ceil((t - r + 2 * p + 1) / s)
    pad = size - math.ceil((t - r + p + 1) / s)
    return size, max(0, pad)  # NTU-60, T=300: (75, 19)


class CoStGcnOracle:
    """CoStGcn frame-by-frame driver (models/base.py:68-122,183-190; models/cost_gcn/cost_gcn.py:30-41).

    ``sd`` uses the *regular* StGcn key layout (``layers.layerK.gcn...``) -- i.e. what
    ``map_state_dict`` (base.py:200-224) maps from.  One call to ``forward_step`` consumes a frame
    (N, C, V, M) and returns logits (N, classes) on the steps where the whole stack emits, else None.
    The temporal average pool is ``AvgPool1d(pool_size, stride 1)`` over the emitted layer-10
    features with a zero-initialised window of pool_size-1 entries and a fixed divisor
    (count_include_pad); it emits from the (pool_size - pool_padding)-th feature on.
    """

    def __init__(self, sd: Dict[str, Tensor], c_in: int = 3, pool_size: int = 75, pool_padding: int = 19):
        self.sd = sd
        self.blocks = [
            CoBlockOracle(sd, f"layers.layer{i + 1}.", st, res, padding=4)
            for i, (_, _, st, res) in enumerate(layer_table(c_in))
        ]
        self.pool_size, self.pool_padding = pool_size, pool_padding
        self.clean_state()

    def clean_state(self):
        for b in self.blocks:
            b.clean_state()
        self.pool: deque = deque(maxlen=self.pool_size)
        self.pool_seen = 0

    def features_step(self, x_t: Tensor) -> Optional[Tensor]:
        """(N, C, V, M) -> layer-10 output (N*M, 256, V) or None."""
        n, c, v, m = x_t.shape
        h = x_t.permute(0, 3, 2, 1).contiguous().view(n, m * v * c)         # base.py:73-75
        h = _bn(h, self.sd, "data_bn.")                                     # base.py:76
        h = h.view(n, m, v, c).permute(0, 1, 3, 2).contiguous().view(n * m, c, v)  # base.py:77-82
        for b in self.blocks:
            h = b.forward_step(h)
            if h is None:
                return None
        return h

    def forward_step(self, x_t: Tensor) -> Optional[Tensor]:
        n, c, v, m = x_t.shape
        h = self.features_step(x_t)
        if h is None:
            return None
        f = h.view(n, m, h.shape[1], v).mean(3).mean(1)                     # base.py:84
        self.pool.append(f)
        self.pool_seen += 1
        if self.pool_seen < self.pool_size - self.pool_padding:
            return None
        pooled = torch.stack(list(self.pool), 0).sum(0) / self.pool_size     # base.py:97
        return F.linear(pooled, self.sd["fc.weight"], self.sd["fc.bias"])   # base.py:99


# --------------------------------------------------------------------------------------------
# Multi-stream fusion + top-k (reference: scripts/multi_stream_eval.py:33-60)
# --------------------------------------------------------------------------------------------
def fuse_preds(preds: Sequence[np.ndarray], method=np.add) -> np.ndarray:
    """``reduce(method, preds[1:], preds[0])`` then ``[:, :, 0]`` for step outputs (multi_stream_eval.py:41,56-57)."""
    from functools import reduce
    out = reduce(method, preds[1:], preds[0])
    return out[:, :, 0] if out.ndim == 3 else out


def topk_accuracies_np(preds: np.ndarray, targets: np.ndarray, ks=(1, 3, 5)):
    """Fraction of samples whose target is among the k best scores (ties: a class only outranks the target when
    it scores strictly higher)."""
    tv = preds[np.arange(len(targets)), targets]
    rank = (preds > tv[:, None]).sum(1)
    return [float((rank < k).mean()) for k in ks]
