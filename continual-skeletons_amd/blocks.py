"""Clip-mode block library: GraphConvolution / TemporalConvolution / SpatioTemporalBlock.

Same constructors, attribute names and ``state_dict`` layout as the reference's
``models/base.py:230-387`` (an unmodified reference state_dict loads with ``strict=True``), but
``forward`` launches the fused gfx950 stage kernels through the C ABI (include/cskel.h) instead of
issuing ~14 ATen ops per block.  Inference only: BatchNorm is folded with its running statistics, so
a module in training mode raises.  The ``nn.Conv2d`` / ``nn.BatchNorm2d`` children are parameter
containers (they define the state_dict keys and initialisation); they are never called.
"""
import math

import numpy as np
import torch
import torch.nn as nn
from torch.nn.modules.batchnorm import _BatchNorm
from torch.nn.modules.conv import _ConvNd

from . import fold, native


# ---- models/utils.py:9-32 -----------------------------------------------------------------------
def init_weights(module_, bs=1):
    """Initialisation rule of the reference (models/utils.py:9-24)."""
    if isinstance(module_, _ConvNd):
        nn.init.constant_(module_.bias, 0)
        if bs == 1:
            nn.init.kaiming_normal_(module_.weight, mode="fan_out")
        else:
            nn.init.normal_(module_.weight, 0, math.sqrt(2.0 / (module_.weight.numel() * bs)))
    elif isinstance(module_, _BatchNorm):
        nn.init.constant_(module_.weight, bs)
        nn.init.constant_(module_.bias, 0)
    elif isinstance(module_, nn.Linear):
        nn.init.normal_(module_.weight, 0, math.sqrt(2.0 / bs))


def zero(x):
    return 0


def unity(x):
    return x


# ---- folded-operand cache -------------------------------------------------------------------------
class _Folded(nn.Module):
    """Mixin: caches the packed device operands and refolds when any parameter/buffer changed
    (load_state_dict, .to(), in-place edits) -- SURVEY 8b 'Folding must be redone if weights are reloaded'.

    Staleness is checked on EVERY call and exactly, but cheaply: the cache keeps a flat snapshot of where the watched
    tensors and sub-modules LIVE (owning ``_parameters`` / ``_buffers`` / ``_modules`` dict + name, with the tensor's
    identity, storage pointer and version counter); re-reading those dict slots costs ~0.15 us each, where walking
    ``parameters()`` / ``buffers()`` cost 35-57 us per module -- 0.94 ms of host time per StGcn forward, more than a
    batch-1 clip takes on the GPU.  A replaced Parameter or buffer, a swapped / added / removed sub-module, an in-place
    edit (version counter) and ``p.data = ...`` (storage pointer) are all seen on the next call."""

    def _watched(self):
        """Modules whose tensors the packed operands are folded from (default: this module and everything below it)."""
        return [self]

    def _snapshot(self):
        tensors, modules, seen = [], [(self._modules, tuple(self._modules.items()))], set()   # own slots: a swapped child
        for root in self._watched():
            for m in root.modules():
                if id(m) in seen:
                    continue
                seen.add(id(m))
                for d in (m._parameters, m._buffers):
                    for name, t in d.items():
                        tensors.append((d, name, t, None if t is None else t.data_ptr(), None if t is None else t._version))
                modules.append((m._modules, tuple(m._modules.items())))
        return tensors, modules

    @staticmethod
    def _stale(snapshot):
        tensors, modules = snapshot
        for d, name, t, ptr, ver in tensors:
            cur = d.get(name)
            if cur is not t or (t is not None and (cur._version != ver or cur.data_ptr() != ptr)):
                return True
        for d, items in modules:
            if len(d) != len(items):
                return True
            for name, child in items:
                if d.get(name) is not child:
                    return True
        return False

    def _fingerprint(self):
        """(storage pointer, version) of every watched tensor -- the slow, walk-everything form (kept for tests and tools)."""
        ts = [t for root in self._watched() for t in list(root.parameters()) + list(root.buffers())]
        return tuple((t.data_ptr(), t._version) for t in ts)

    def _packed_ops(self, device):
        cache = self.__dict__.get("_fold_cache")
        device = str(device)
        if cache is None or cache[0] != device or self._stale(cache[2]):
            # "*_host" entries are read by the C ABI on the host (e.g. ell_cnt); everything else lives in HBM
            ops = {k: (v.to(device) if isinstance(v, torch.Tensor) and not k.endswith("_host") else v)
                   for k, v in self._fold().items()}
            self.__dict__["_fold_cache"] = cache = (device, ops, self._snapshot())
        return cache[1]

    def refold(self):
        self.__dict__.pop("_fold_cache", None)

    def _require_eval(self):
        if self.training:
            raise RuntimeError(
                f"{type(self).__name__}: the MI355X path is inference-only (BatchNorm folded with running "
                "statistics); call .eval() first"
            )


PRECISIONS = ("f32", "bf16x3")


def set_precision(module: nn.Module, precision: str = "f32") -> nn.Module:
    """Select the arithmetic of the temporal-conv kernels of ``module`` and every block below it.

    "f32" (default): exact fp32 (v_mfma_f32_32x32x2_f32) -- bit for bit an fmaf chain, the reference's own arithmetic.
    "bf16x3" (opt-in): fp32-GRADE -- every operand split into three bf16 pieces, six piece products per fp32 product on
    the bf16 matrix pipe, fp32 accumulation (csrc/tcn_split.hip).  Results differ from "f32" by a few 1e-6 on O(1)
    activations (tests/test_gpu_precision_modes.py records the measured error); it is never reported as fp32.  Applies
    to the 9 x 1 temporal conv + 1 x 1 residual conv of SpatioTemporalBlock / CoSpatioTemporalBlock (70 % of the FLOPs);
    and to the channel-mixing GEMM of plain ``GraphConvolution`` modules with 128-row tiles (the adjacency aggregation stays
    exact fp32; the 64-channel layers and A-GCN's adaptive graph conv keep the exact kernels)."""
    if precision not in PRECISIONS:
        raise ValueError(f"precision must be one of {PRECISIONS}, got {precision!r}")
    blocks_ = [m for m in module.modules() if isinstance(m, SpatioTemporalBlock)]
    if not blocks_:
        raise ValueError("no SpatioTemporalBlock below this module: nothing to set")
    # validate EVERY block before touching any: a refusal must not leave a half-switched (mixed-precision) model behind
    for m in blocks_:
        if precision != "f32" and not (m._native_tail and m.tcn.kernel_size == 9):
            raise NotImplementedError("bf16x3 is built for blocks with the native 9 x 1 temporal conv; the model is unchanged")
    for m in module.modules():
        if isinstance(m, SpatioTemporalBlock) or type(m) is GraphConvolution:
            m.precision = precision
            m.refold()
    return module


class SplitScratch:
    """Partial-sum buffers of the split-K launches of the clip latency mode.  Owned by the module tree that
    ``set_clip_latency_mode`` was called on (every block below shares one instance; a block whose ``clip_split_k`` was set
    by hand gets one of its own) -- no module-global state.  One buffer per (device, HIP stream): launches on a stream are
    ordered, so all layers reuse it.  A request larger than the current buffer allocates a new one and KEEPS the superseded
    buffers alive: a captured hipGraph may hold their addresses.  ``release()`` drops them all (call it when no graph that
    was captured in latency mode will be replayed again; ``set_clip_latency_mode(module, 0)`` does)."""

    def __init__(self):
        self._bufs = {}          # (device, stream handle) -> [buffers], newest (largest) last

    def get(self, device, floats: int) -> torch.Tensor:
        key = (str(device), torch.cuda.current_stream(device).cuda_stream)
        bufs = self._bufs.setdefault(key, [])
        if not bufs or bufs[-1].numel() < floats:
            bufs.append(torch.empty(int(floats), device=device, dtype=torch.float32))
        return bufs[-1]

    def release(self):
        self._bufs.clear()

    def nbytes(self) -> int:
        return 4 * sum(b.numel() for bufs in self._bufs.values() for b in bufs)


def _scratch_of(module) -> SplitScratch:
    sc = module.__dict__.get("_split_scratch")
    if sc is None:
        sc = module.__dict__["_split_scratch"] = SplitScratch()
    return sc


CLIP_SPLIT_MAX_SEQUENCES = 6   # N * M up to which the split forward is the faster one (NTU batch 3); measured: profiles/HISTORY.md r5 #4, r6 #18


def set_clip_latency_mode(module: nn.Module, split_k: int = 4, gcn_split_k: int = None, max_sequences: int = CLIP_SPLIT_MAX_SEQUENCES) -> nn.Module:
    """Small-batch clip inference (the reference's own CPU protocol is batch 1, scripts/benchmark_all_ntu60.py:17).

    At a few clips a stage launch is a few dozen tiles on 256 CUs, each walking its whole K loop alone (a 256-channel
    temporal conv: 2 x 15 x 2 = 60 workgroups x 288 K-chunks at batch 1).  With ``split_k`` > 1 every 9 x 1 temporal conv
    and every plain graph conv (>= 16 input channels) below ``module`` cuts its K loop into ``split_k`` channel ranges
    computed by separate workgroups (csk_tcn_stage_splitk_f32 / csk_gcn_stage_splitk_f32) and summed in split order by a
    second launch.  The factor is a function of ``split_k`` and the layer's channel count ONLY, and it applies to forwards
    of at most ``max_sequences`` skeleton sequences (N * M; default 6 = NTU batch 3) -- larger forwards run the default
    kernels, which are the faster ones from there on (batch 8: 2.9 ms against 3.5 ms split), so the mode is never slower
    than the default.  A clip's logits are therefore bitwise the same alone and in any batch on the same side of that
    bound; across it (and against the default mode) they differ by the summation order (parity with the oracle within the
    same 1e-4).  ``split_k`` <= 1 switches it off.  ``gcn_split_k`` (default: ``split_k``) sets the graph convs' factor
    separately."""
    gcn_split_k = split_k if gcn_split_k is None else gcn_split_k
    for v in (split_k, gcn_split_k):
        if not isinstance(v, int) or v < 0 or v > 32:
            raise ValueError("split_k must be an integer in [0, 32]")
    targets = [m for m in module.modules() if isinstance(m, SpatioTemporalBlock) or type(m) is GraphConvolution]
    if not targets:
        raise ValueError("no SpatioTemporalBlock / GraphConvolution below this module: nothing to set")
    if not isinstance(max_sequences, int) or max_sequences < 1:
        raise ValueError("max_sequences must be a positive integer")
    # one partial-sum scratch for the whole tree (non-persistent, not part of state_dict); switching the mode off releases it
    old = {id(sc): sc for sc in (m.__dict__.get("_split_scratch") for m in targets) if sc is not None}
    for sc in old.values():
        sc.release()
    shared = SplitScratch() if max(split_k, gcn_split_k) > 1 else None
    for m in targets:
        m.clip_split_k = split_k if isinstance(m, SpatioTemporalBlock) else gcn_split_k
        m.clip_split_max_seq = int(max_sequences)
        m.__dict__["_split_scratch"] = shared
    return module


def _check_input(x, channels, name):
    native.require_device_f32(x, name)
    if x.dim() != 4 or x.shape[1] != channels:
        raise RuntimeError(f"{name} must be (N, {channels}, T, V), got {tuple(x.shape)}")


# ---- models/base.py:230-270 -----------------------------------------------------------------------
class GraphConvolution(_Folded):
    """K=3-partition adjacency aggregation + three 1x1 convs + BN + (identity | conv+BN) residual + ReLU."""

    def __init__(self, in_channels, out_channels, A, bn_momentum=0.1, *args, **kwargs):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.graph_attn = nn.Parameter(torch.ones(A.shape, dtype=torch.float32))
        self.A = nn.Parameter(torch.from_numpy(np.asarray(A).astype(np.float32)), requires_grad=False)
        self.num_subset = 3
        self.g_conv = nn.ModuleList(nn.Conv2d(in_channels, out_channels, 1) for _ in range(self.num_subset))
        for conv in self.g_conv:
            init_weights(conv, bs=self.num_subset)
        if in_channels != out_channels:
            self.gcn_residual = nn.Sequential(
                nn.Conv2d(in_channels, out_channels, 1), nn.BatchNorm2d(out_channels, momentum=bn_momentum)
            )
            init_weights(self.gcn_residual[0], bs=1)
            init_weights(self.gcn_residual[1], bs=1)
        else:
            self.gcn_residual = unity
        self.bn = nn.BatchNorm2d(out_channels, momentum=bn_momentum)
        init_weights(self.bn, bs=1e-6)
        self.relu = nn.ReLU()

    precision = "f32"      # or "bf16x3" (opt-in, set_precision): arithmetic of the channel-mixing GEMM of the clip forward

    def _fold(self):
        return fold.fold_graph_conv(self.state_dict(), split=self.precision == "bf16x3")

    def _split_applies(self, ops):
        """csk_gcn_stage_bf16x3 is built for skeleton-sparse graphs and 128-row tiles; other shapes (the 64-channel
        layers) keep the exact kernel in that mode."""
        cnt = ops["ell_cnt_host"]
        return (self.precision == "bf16x3" and ops["w_split"] is not None and self.out_channels % 128 == 0
                and int(cnt[0]) <= 1 and int(cnt[1]) <= 1 and int(cnt[2]) <= 4)

    def forward(self, x):
        self._require_eval()
        _check_input(x, self.in_channels, "GraphConvolution input")
        ops = self._packed_ops(x.device)
        n, c, t, v = x.shape
        if v != ops["V"]:
            raise RuntimeError(f"input has V={v} joints, adjacency has {ops['V']}")
        y = torch.empty((n, self.out_channels, t, v), device=x.device, dtype=torch.float32)
        if self._split_applies(ops):
            rc = native.lib().csk_gcn_stage_bf16x3(
                native.ptr(x), native.ptr(y), native.ptr(ops["w_split"]), native.ptr(ops["w_res_split"]), native.ptr(ops["bias"]),
                native.ptr(ops["ell_src"]), native.ptr(ops["ell_val"]), native.ptr(ops["ell_cnt_host"]), ops["ell_w"], n, c,
                self.out_channels, t, v, ops["res_mode"], native.stream_of(x))
            native.check(rc, "csk_gcn_stage_bf16x3")
            return y
        ks = self._clip_ksplit() if n <= self.clip_split_max_seq else 1
        if ks > 1:
            self.stage(x, y, n_seg=n, frames=t, x_strides=(c * t * v, t * v), y_strides=(self.out_channels * t * v, t * v),
                       ksplit=ks, partial=_scratch_of(self).get(x.device, ks * n * self.out_channels * t * v))
            return y
        gcn_stage(x, y, ops, n_seg=n, frames=t, x_strides=(c * t * v, t * v), y_strides=(self.out_channels * t * v, t * v))
        return y

    clip_split_k = 0       # set_clip_latency_mode: channel ranges per tile of the clip forward (0 / 1: off)
    clip_split_max_seq = CLIP_SPLIT_MAX_SEQUENCES   # ... for forwards of at most this many sequences

    def _clip_ksplit(self) -> int:
        """Split factor of the clip forward: min(clip_split_k, 8-channel chunks of the K loop) for plain graph convs with
        >= 16 input channels -- a function of (clip_split_k, C_in) only."""
        if self.clip_split_k <= 1 or type(self) is not GraphConvolution or self.in_channels < 16:
            return 1
        return max(1, min(self.clip_split_k, -(-self.in_channels // 8)))

    def stage(self, x, y, n_seg, frames, x_strides, y_strides, ksplit=1, partial=None):
        """Launch on explicit views/strides (used by the continual engine on its channel-major rings).  ``ksplit`` > 1
        (latency mode): the K loop is split over workgroups through ``partial`` (csk_gcn_stage_splitk_f32)."""
        ops = self._packed_ops(x.device)
        if ksplit > 1:
            rc = native.lib().csk_gcn_stage_splitk_f32(
                native.ptr(x), native.ptr(y), native.ptr(ops["w"]), native.ptr(ops["bias"]), native.ptr(ops["ell_src"]),
                native.ptr(ops["ell_val"]), native.ptr(ops["ell_cnt_host"]), ops["ell_w"], n_seg, ops["c_in"], ops["c_out"], frames,
                ops["V"], x_strides[0], x_strides[1], y_strides[0], y_strides[1], ops["res_mode"], ksplit, native.ptr(partial),
                native.stream_of(x))
            native.check(rc, "csk_gcn_stage_splitk_f32")
            return
        gcn_stage(x, y, ops, n_seg=n_seg, frames=frames, x_strides=x_strides, y_strides=y_strides)


def gcn_stage(x, y, ops, n_seg, frames, x_strides, y_strides, adj_seg_stride=0, adj_per_frame=0):
    rc = native.lib().csk_gcn_stage_f32(
        native.ptr(x), native.ptr(y), native.ptr(ops["w"]), native.ptr(ops["bias"]),
        native.ptr(ops["ell_src"]), native.ptr(ops["ell_val"]), native.ptr(ops["ell_cnt_host"]),
        ops["ell_w"], adj_seg_stride, adj_per_frame, n_seg, ops["c_in"], ops["c_out"], frames, ops["V"],
        x_strides[0], x_strides[1], y_strides[0], y_strides[1], ops["res_mode"], native.stream_of(x),
    )
    native.check(rc, "csk_gcn_stage_f32")


# ---- models/base.py:279-304 -----------------------------------------------------------------------
class TemporalConvolution(_Folded):
    """Conv2d (k,1) stride (s,1) pad (p,0) + BatchNorm2d."""

    def __init__(self, in_channels, out_channels, kernel_size=9, stride=1, padding=4):
        super().__init__()
        self.padding = padding
        self.stride = stride
        self.kernel_size = kernel_size
        self.t_conv = nn.Conv2d(in_channels, out_channels, kernel_size=(kernel_size, 1), padding=(padding, 0),
                                stride=(stride, 1))
        self.bn = nn.BatchNorm2d(out_channels)
        init_weights(self.t_conv, bs=1)
        init_weights(self.bn, bs=1)

    def _fold(self):
        f = fold.fold_temporal_conv(self.state_dict())
        f["bias"] = fold.pad_vec(f["bias"])
        return f

    def forward(self, x):
        self._require_eval()
        _check_input(x, self.t_conv.in_channels, "TemporalConvolution input")
        ops = self._packed_ops(x.device)
        return tcn_stage(x, ops["w"], ops["bias"], ops["c_out"], ops["k"], self.stride, self.padding, relu=False)


def tcn_stage(y, w, bias, c_out, k, stride, pad, relu=True, res_mode=0, x_res=None, w_res=None, res_off=0, out=None,
              split=False, ksplit=1, scratch=None):
    """csk_tcn_stage_f32, or with split=True csk_tcn_stage_bf16x3 (w / w_res are then the split operand images), or with
    ksplit > 1 csk_tcn_stage_splitk_f32 (clip latency mode: K loop cut into channel ranges, partial sums in split order)."""
    n, c, t_in, v = y.shape
    if t_in + 2 * pad < k:
        raise RuntimeError(f"temporal extent {t_in} (+2*{pad}) shorter than kernel {k}")
    t_out = (t_in + 2 * pad - k) // stride + 1
    if out is None:
        out = torch.empty((n, c_out, t_out, v), device=y.device, dtype=torch.float32)
    elif tuple(out.shape) != (n, c_out, t_out, v) or not out.is_contiguous() or out.dtype != torch.float32 or out.device != y.device:
        raise RuntimeError(f"out must be a contiguous float32 {(n, c_out, t_out, v)} tensor on {y.device}")
    c_res, t_res = (x_res.shape[1], x_res.shape[2]) if x_res is not None else (0, 0)
    if ksplit > 1 and not split:
        if scratch is None:
            raise RuntimeError("tcn_stage: ksplit > 1 needs the owning module's SplitScratch")
        part = scratch.get(y.device, ksplit * n * c_out * t_out * v)
        rc = native.lib().csk_tcn_stage_splitk_f32(
            native.ptr(y), native.ptr(w), native.ptr(x_res), native.ptr(w_res), native.ptr(bias), native.ptr(out),
            n, c, c_out, t_in, v, k, stride, pad, res_mode, c_res, t_res, res_off, int(relu), ksplit, native.ptr(part),
            native.stream_of(y))
        native.check(rc, "csk_tcn_stage_splitk_f32")
        return out
    fn = native.lib().csk_tcn_stage_bf16x3 if split else native.lib().csk_tcn_stage_f32
    rc = fn(native.ptr(y), native.ptr(w), native.ptr(x_res), native.ptr(w_res), native.ptr(bias), native.ptr(out),
            n, c, c_out, t_in, v, k, stride, pad, res_mode, c_res, t_res, res_off, int(relu), native.stream_of(y))
    native.check(rc, "csk_tcn_stage_bf16x3" if split else "csk_tcn_stage_f32")
    return out


# ---- models/base.py:337-387 -----------------------------------------------------------------------
class SpatioTemporalBlock(_Folded):
    """ReLU(tcn(gcn(x)) + residual(x)); residual in {zero, identity, conv1x1(stride)+BN}; optional
    centred residual shrink when ``temporal_padding`` is given (the "*" variants)."""

    def __init__(self, in_channels, out_channels, A, stride=1, residual=True, temporal_kernel_size=9,
                 temporal_padding=-1, GraphConv=GraphConvolution, TempConv=TemporalConvolution):
        super().__init__()
        equal_padding = int((temporal_kernel_size - 1) / 2)
        if temporal_padding < 0:
            temporal_padding = equal_padding
            self.residual_shrink = None
        else:
            assert temporal_padding <= equal_padding
            self.residual_shrink = equal_padding - temporal_padding
        self.stride = stride
        self.gcn = GraphConv(in_channels, out_channels, A)
        self.tcn = TempConv(out_channels, out_channels, stride=stride, kernel_size=temporal_kernel_size,
                            padding=temporal_padding)
        self.relu = nn.ReLU()
        if not residual:
            self.residual = zero
        elif (in_channels == out_channels) and (stride == 1):
            self.residual = unity
        else:
            self.residual = TempConv(in_channels, out_channels, kernel_size=1, stride=stride, padding=0)
        self._native_tail = isinstance(self.tcn, TemporalConvolution) and (
            self.residual in (zero, unity) or isinstance(self.residual, TemporalConvolution)
        )

    precision = "f32"      # or "bf16x3" (opt-in, set_precision): arithmetic of the temporal conv / residual conv kernels
    clip_split_k = 0       # set_clip_latency_mode: channel ranges per tile of the clip forward's temporal conv (0 / 1: off)
    clip_split_max_seq = CLIP_SPLIT_MAX_SEQUENCES   # ... for forwards of at most this many sequences

    def _watched(self):    # the tail's operands are folded from the temporal conv and the residual conv; the graph conv keeps its own cache
        return [m for m in (self.tcn, self.residual) if isinstance(m, nn.Module)]

    def _fold(self):
        sd = self.state_dict()
        return fold.fold_block_tail(sd, "", has_conv_residual=isinstance(self.residual, TemporalConvolution),
                                    split=self.precision == "bf16x3", stride=self.stride)

    def forward(self, x, out=None):
        """``out`` (optional, native tail only): preallocated (N, C_out, T_out, V) tensor to write into."""
        self._require_eval()
        native.require_device_f32(x, "SpatioTemporalBlock input")
        if self._few_channels_fusable(x):
            return self._forward_few_channels(x, out)
        y = self.gcn(x)                                   # GCN stage kernel (or any user GraphConv module)
        if not self._native_tail:
            # foreign TempConv / residual modules: compose exactly as models/base.py:376-387
            z = self.tcn(y)
            r = self.residual(x[:, :, self.residual_shrink:-self.residual_shrink] if self.residual_shrink else x)
            return self.relu(z + r)
        ops = self._packed_ops(x.device)
        shrink = self.residual_shrink or 0
        if self.residual is zero:
            mode, xr = 0, None
        elif self.residual is unity:
            mode, xr = 1, x
        else:
            mode, xr = 2, x
        if self.precision == "bf16x3":
            return tcn_stage(y, ops["w_split"], ops["bias"], ops["c_out"], ops["k"], self.stride, self.tcn.padding, relu=True,
                             res_mode=mode, x_res=xr, w_res=ops["w_res_split"], res_off=shrink, out=out, split=True)
        ks = max(1, min(self.clip_split_k, -(-ops["c"] // 8))) if (self.clip_split_k > 1 and ops["k"] == 9 and y.shape[0] <= self.clip_split_max_seq) else 1
        return tcn_stage(y, ops["w"], ops["bias"], ops["c_out"], ops["k"], self.stride, self.tcn.padding, relu=True,
                         res_mode=mode, x_res=xr, w_res=ops["w_res"], res_off=shrink, out=out, ksplit=ks,
                         scratch=_scratch_of(self) if ks > 1 else None)


def _few_channels_fusable(self, x) -> bool:
    """Layer 1 of the stacks (st_gcn.py:30): a plain GraphConvolution with <= 4 input channels and a conv gcn_residual, the native
    9-tap / stride-1 / pad-4 temporal tail, no block residual, exact fp32, no split-K request -> csk_block_few_channels_f32."""
    if not self.fuse_few_channels or type(self.gcn) is not GraphConvolution or not self._native_tail:
        return False
    if self.gcn.in_channels > 4 or self.residual is not zero or self.stride != 1 or self.precision != "f32" or self.residual_shrink:
        return False
    if x.dim() != 4 or x.shape[3] not in (18, 25) or self.gcn.precision != "f32":
        return False
    if self.clip_split_k > 1 and x.shape[0] <= self.clip_split_max_seq:
        return False
    g, t = self.gcn._packed_ops(x.device), self._packed_ops(x.device)
    return (t["k"] == 9 and self.tcn.padding == 4 and g["res_mode"] == 2 and g["c_out"] % 8 == 0 and g["V"] == x.shape[3]
            and max(int(v) for v in g["ell_cnt_host"][:2]) <= 1 and int(g["ell_cnt_host"][2]) <= 4)


def _forward_few_channels(self, x, out=None):
    _check_input(x, self.gcn.in_channels, "SpatioTemporalBlock input")
    g, t = self.gcn._packed_ops(x.device), self._packed_ops(x.device)
    n, c, tt, v = x.shape
    if out is None:
        out = torch.empty((n, t["c_out"], tt, v), device=x.device, dtype=torch.float32)
    rc = native.lib().csk_block_few_channels_f32(
        native.ptr(x.contiguous()), native.ptr(g["w"]), native.ptr(g["bias"]), native.ptr(g["ell_src"]), native.ptr(g["ell_val"]),
        native.ptr(g["ell_cnt_host"]), g["ell_w"], native.ptr(t["w"]), native.ptr(t["bias"]), native.ptr(out), n, c, g["c_out"],
        t["c_out"], tt, v, self.tcn.padding, native.stream_of(x))
    native.check(rc, "csk_block_few_channels_f32")
    return out


# Opt-in (default: the two launches).  Measured at batch 256 (tools/ab_first_block.py): block 1 fused 2.61-2.63 ms against 2.58-2.59 ms
# for its two launches, the forward 69.45 against 69.36 ms -- the graph-conv launch it saves (0.36 ms, 2 GB of HBM traffic) comes
# back as the serial y phase of every chunk and the recomputed halo; bit for bit the same output either way.
SpatioTemporalBlock.fuse_few_channels = False
SpatioTemporalBlock._few_channels_fusable = _few_channels_fusable
SpatioTemporalBlock._forward_few_channels = _forward_few_channels


def tcn_step_launch(*args):
    """csk_tcn_step_f32 launch (module-level so that bench.py can time it with HIP events)."""
    native.check(native.lib().csk_tcn_step_f32(*args), "csk_tcn_step_f32")
