"""Continual (frame-by-frame) block library and the CoST-GCN step driver.

Counterpart of the reference's ``CoGraphConvolution`` / ``CoTemporalConvolution`` /
``CoSpatioTemporalBlock`` (models/base.py:273-276, 307-334, 390-446), ``CoModelBase`` (base.py:19-227)
and ``CoStGcn`` (models/cost_gcn/cost_gcn.py).  In the reference the step arithmetic lives in the
third-party ``continual-inference`` package (per-module Python-side buffers, one small ATen op per
module per frame); here every block owns a slice of a persistent HBM state slab and a step is two
kernel launches (csk_gcn_stage_f32 on the new frame, csk_tcn_step_f32 over the ring).

Protocol (anchored on the reference's tests, see oracle/stgcn_oracle.py:CoBlockOracle):
  * ``forward_step(x_t)`` -> output frame or ``None``; nothing is emitted during the first
    ``delay = k-1-padding`` steps, and with temporal stride S only every S-th step emits;
  * the emission of step s equals the clip block's output at t = (s - delay) / S;
  * ``forward_steps(x, pad_end)``: all frames, optionally flushed with ``padding`` zero post-GCN frames;
  * ``clean_state()`` zeroes the window (zero state == the clip conv's left zero padding).
State layout (channel-major, see include/cskel.h): per block a y ring [8 + max_in][C_out][P] and an output ring
[4 + max_in of the next block][C_out][P] (max_in = frames one launch can receive = 8 / cumulative stride); the output
ring of block l is the input/residual history of block l+1, so the residual FIFO (``co.Delay``) costs no copy.  ``engine_advance`` consumes up to 8 frames per call (4 = one
stride cycle of the 10-block stack) with one GCN launch and one multi-emission TCN launch per block,
which is what fills the GPU at ~1000 streams; per-frame stepping is the same code with r = 1.
"""
import ctypes
import math
from collections import OrderedDict
from typing import Optional

import torch
import torch.nn as nn

from . import blocks, fold, native
from .blocks import GraphConvolution, SpatioTemporalBlock, TemporalConvolution, _Folded, init_weights, unity, zero
from .models import layer_table

MAX_CYCLE = 8


def y_slots(max_in: int) -> int:
    """Depth of a post-GCN ring: the k-1 = 8 window frames of co.Conv2d + the frames one launch can receive
    (include/cskel.h: CSK_CO_Y_SLOTS)."""
    return 8 + max_in


def in_slots(max_in: int) -> int:
    """Depth of an input / output history ring: residual lag (k-1)/2 = 4 (co.Delay) + the frames one launch of the
    CONSUMING block can receive (include/cskel.h: CSK_CO_IN_SLOTS)."""
    return 4 + max_in


def _round4(n: int) -> int:
    return (n + 3) // 4 * 4


def CoGraphConvolution(in_channels, out_channels, A, bn_momentum=0.1):
    """models/base.py:273-276 -- the per-frame graph conv is stateless, so it is the same module."""
    return GraphConvolution(in_channels, out_channels, A, bn_momentum)


class CoTemporalConvolution(TemporalConvolution):
    """models/base.py:307-334: (k,1) conv + BN with a (k-1)-frame window.  ``padding="equal"`` -> (k-1)/2.
    Note the reference's argument order (kernel_size, padding, stride) differs from TemporalConvolution's."""

    def __init__(self, in_channels, out_channels, kernel_size=9, padding=0, stride=1):
        if padding == "equal":
            padding = int((kernel_size - 1) / 2)
        super().__init__(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding)
        self.receptive_field = kernel_size
        self.delay = kernel_size - 1 - padding
        self._ring = None
        self._s = 0

    # -- continual interface on (N, C, V) frames -------------------------------------------------
    def clean_state(self):
        self._ring, self._s = None, 0

    def _state(self, n, v, device):
        p = _round4(n * v)
        c = self.t_conv.in_channels
        if self._ring is None or self._ring.shape != (self.kernel_size, c, p) or self._ring.device != device:
            self._ring = torch.zeros((self.kernel_size, c, p), device=device, dtype=torch.float32)
            self._s = 0
        return p

    def _emit(self, n, v, p):
        ops = self._packed_ops(self._ring.device)
        out = torch.empty((ops["c_out"], p), device=self._ring.device, dtype=torch.float32)
        rc = native.lib().csk_tcn_step_f32(
            native.ptr(self._ring), self.kernel_size, self._s % self.kernel_size, 0, 1, native.ptr(ops["w"]),
            None, 0, 0, 0, None, native.ptr(ops["bias"]), native.ptr(out), 1, 0,
            ops["c_in"], ops["c_out"], p, self.kernel_size, 0, 0, 0, 1, None, native.stream_of(out))
        native.check(rc, "csk_tcn_step_f32")
        return out[:, : n * v].view(-1, n, v).permute(1, 0, 2).contiguous()

    def forward_step(self, x_t, update_state=True):
        """One frame.  ``update_state=False`` computes the step without advancing: the frame lands in the ring slot
        of the frame that has just left the window, so putting the counter back is all there is to undo."""
        self._require_eval()
        native.require_device_f32(x_t, "CoTemporalConvolution frame")
        n, c, v = x_t.shape
        p = self._state(n, v, x_t.device)
        self._ring[self._s % self.kernel_size, :, : n * v] = x_t.permute(1, 0, 2).reshape(c, n * v)
        s = self._s
        out = None
        if s >= self.delay and (s - self.delay) % self.stride == 0:
            out = self._emit(n, v, p)
        if update_state:
            self._s += 1
        return out

    def forward_steps(self, x, pad_end=False, update_state=True):
        n, c, t, v = x.shape
        if not update_state:                       # several frames overwrite live window slots: keep a copy
            self._state(n, v, x.device)
            keep = (self._ring.clone(), self._s)
            try:
                return self.forward_steps(x, pad_end, True)
            finally:
                self._ring.copy_(keep[0])
                self._s = keep[1]
        outs = [o for o in (self.forward_step(x[:, :, i].contiguous()) for i in range(t)) if o is not None]
        if pad_end:
            p = self._state(n, v, x.device)
            for _ in range(self.padding):
                self._ring[self._s % self.kernel_size].zero_()
                if self._s >= self.delay and (self._s - self.delay) % self.stride == 0:
                    outs.append(self._emit(n, v, p))
                self._s += 1
        return torch.stack(outs, dim=2)


class _BlockState:
    """Slice of the state slab owned by one block: y ring, output ring, (optionally own) input ring, counters."""

    def __init__(self, c_in, c_out, k, p, device, xin=None, ksplit=1, max_emit=MAX_CYCLE, scratch=None, max_in=MAX_CYCLE,
                 out_slots=None, gcn_ksplit=1):
        self.p = p
        self.ksplit = ksplit
        self.gcn_ksplit = gcn_ksplit      # split-K of the graph conv (latency mode); shares the partial-sum buffer
        # split-K scratch of the TCN step: raw partial sums of the emissions ONE launch can produce (max_emit: MAX_CYCLE
        # for a stand-alone block, MAX_CYCLE / cumulative stride inside a stack).  Launches are stream-ordered, so a
        # stack shares one scratch buffer (``scratch``, sized by its largest user) instead of one per block.
        self.max_emit = max_emit
        self.owns_partial = (ksplit > 1 or gcn_ksplit > 1) and scratch is None
        slabs = max(max_emit * ksplit if ksplit > 1 else 0, max_in * gcn_ksplit if gcn_ksplit > 1 else 0)   # [c_out][p] each
        need = slabs * c_out * p
        if slabs == 0:
            self.partial = None
        elif scratch is not None:
            if scratch.numel() < need:
                raise ValueError(f"shared split-K scratch holds {scratch.numel()} floats, block needs {need}")
            self.partial = scratch[:need].view(slabs, c_out, p)
        else:
            self.partial = torch.empty((slabs, c_out, p), device=device, dtype=torch.float32)
        # ring depths from what ONE launch can receive / emit (y_slots / in_slots above); a stand-alone block keeps an
        # output ring deep enough for max_emit emissions (and the 4 a fused cycle writes)
        self.max_in = max_in
        self.y = torch.zeros((y_slots(max_in), c_out, p), device=device, dtype=torch.float32)
        self.out = torch.zeros((out_slots or max(4, max_emit), c_out, p), device=device, dtype=torch.float32)
        self.owns_xin = xin is None
        self.xin = torch.zeros((in_slots(max_in), c_in, p), device=device, dtype=torch.float32) if xin is None else xin
        if self.xin.shape[0] < in_slots(max_in):
            raise ValueError(f"input ring of {self.xin.shape[0]} slots is too shallow for launches of {max_in} frames")
        self.s = 0      # frames received
        self.e = 0      # frames emitted

    def zero_(self):
        self.y.zero_()
        self.out.zero_()
        if self.owns_xin:
            self.xin.zero_()
        self.s = self.e = 0

    def nbytes(self):
        """Persistent state of this block (rings); the split-K scratch is reported by scratch_bytes()."""
        return 4 * (self.y.numel() + self.out.numel() + (self.xin.numel() if self.owns_xin else 0))

    def scratch_bytes(self):
        return 4 * self.partial.numel() if self.owns_partial else 0


class CoSpatioTemporalBlock(SpatioTemporalBlock):
    """models/base.py:390-446.  Same call signature as the reference's factory function; ``forward`` is the
    clip computation (== SpatioTemporalBlock with ``temporal_padding=padding``), ``forward_step`` /
    ``forward_steps`` run on the persistent state.

    state_dict layout = the reference's container layout (tests/test_cost_gcn.py:97-98,145,193-198):
    no residual -> ``gcn.* / tcn.*``; identity -> ``0.1.gcn.* / 0.1.tcn.*``; conv residual ->
    ``0.0.residual.*`` + ``0.1.gcn.* / 0.1.tcn.*``.  The plain SpatioTemporalBlock layout also loads.
    """

    def __init__(self, in_channels, out_channels, A, stride=1, residual=True, window_size=1, padding=0,
                 CoGraphConv=CoGraphConvolution, CoTempConv=None):
        if padding == "equal":
            padding = 4
        window_size = int(window_size)  # unused by the reference as well (base.py:401)

        def graph_conv(ci, co, a):
            return CoGraphConv(ci, co, a, bn_momentum=0.1)

        def temp_conv(ci, co, kernel_size=9, stride=1, padding=0):
            if CoTempConv is None:
                return TemporalConvolution(ci, co, kernel_size=kernel_size, stride=stride, padding=padding)
            return CoTempConv(ci, co, kernel_size=kernel_size, padding=padding, stride=stride)

        super().__init__(in_channels, out_channels, A, stride=stride, residual=residual, temporal_kernel_size=9,
                         temporal_padding=padding, GraphConv=graph_conv, TempConv=temp_conv)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = 9
        self.padding = padding
        self.receptive_field = self.kernel_size
        self.delay = self.kernel_size - 1 - padding
        self.kind = "none" if self.residual is zero else ("identity" if self.residual is unity else "conv")
        self._prefix_map = {"none": {}, "identity": {"gcn.": "0.1.gcn.", "tcn.": "0.1.tcn."},
                            "conv": {"gcn.": "0.1.gcn.", "tcn.": "0.1.tcn.", "residual.": "0.0.residual."}}[self.kind]
        self._state: Optional[_BlockState] = None
        self._register_state_dict_hook(CoSpatioTemporalBlock._to_co_keys)
        self._register_load_state_dict_pre_hook(self._from_co_keys)
        if not self._native_tail:
            raise NotImplementedError("continual blocks need the native TemporalConvolution (the ring-buffer step kernel)")
        # self.gcn may be any per-frame graph-conv module (models/base.py:273-276 applies it frame by frame): native
        # ones bring a ``stage`` method on the channel-major state layout, foreign ones go through _foreign_gcn_stage

    # ---- state_dict key layout -------------------------------------------------------------------
    @staticmethod
    def _to_co_keys(module, state_dict, prefix, local_metadata):
        for plain, co in module._prefix_map.items():
            for k in [k for k in state_dict if k.startswith(prefix + plain)]:
                state_dict[prefix + co + k[len(prefix + plain):]] = state_dict.pop(k)
        return state_dict

    def _from_co_keys(self, state_dict, prefix, *args):
        for plain, co in self._prefix_map.items():
            for k in [k for k in state_dict if k.startswith(prefix + co)]:
                state_dict[prefix + plain + k[len(prefix + co):]] = state_dict.pop(k)

    def _fold(self):                      # fold from the PLAIN layout whatever state_dict() emits
        sd = {}
        for k, v in nn.Module.state_dict(self).items():
            for plain, co in self._prefix_map.items():
                if k.startswith(co):
                    k = plain + k[len(co):]
                    break
            sd[k] = v
        return fold.fold_block_tail(sd, "", has_conv_residual=self.kind == "conv", split=self.precision == "bf16x3", stride=self.stride)

    # ---- persistent state --------------------------------------------------------------------------
    def bind_state(self, p: int, device, xin: Optional[torch.Tensor] = None, max_emit: int = MAX_CYCLE,
                   scratch: Optional[torch.Tensor] = None, max_in: int = MAX_CYCLE, out_slots: Optional[int] = None) -> _BlockState:
        """(Re)allocate this block's slab slice for P positions; ``xin`` = upstream block's output ring; ``max_in`` /
        ``max_emit`` = frames one launch of this block can receive / emit; ``out_slots`` = depth of the output ring (what
        the consuming block needs as its input history); ``scratch`` = split-K scratch shared with the other blocks."""
        self._state = _BlockState(self.in_channels, self.out_channels, self.kernel_size, p, device, xin,
                                  ksplit=self._pick_ksplit(p), max_emit=max_emit, scratch=scratch, max_in=max_in,
                                  out_slots=out_slots, gcn_ksplit=self._pick_gcn_ksplit(p))
        return self._state

    def scratch_floats(self, p: int, max_emit: int = MAX_CYCLE, max_in: int = MAX_CYCLE) -> int:
        """Split-K scratch this block needs for launches of up to ``max_in`` received frames / ``max_emit`` emissions
        (0 without split-K)."""
        ks, gks = self._pick_ksplit(p), self._pick_gcn_ksplit(p)
        return max(max_emit * ks if ks > 1 else 0, max_in * gks if gks > 1 else 0) * self.out_channels * p

    split_k = 0     # 0: no split-K; n > 1: latency mode -- up to n channel ranges per tile when a launch is too small

    THROUGHPUT_SPLIT_K = 1   # split-K of the 256-channel blocks in the default mode (see _pick_ksplit)

    def _pick_ksplit(self, p: int) -> int:
        """Split-K factor of this block's TCN step (csk_tcn_step_f32 ``ksplit``) -- a function of (C_out, split_k) ONLY: a
        stream's results never depend on how many streams share the slab (``p`` is not used).
        * Default mode: no split.  (Rounds 2-5 cut the 2304-deep K loop of the C_out >= 256 blocks into 3 channel ranges to
          pack 800 tiles of 128 x 128 onto 512 resident slots; the slot-balanced tiles of csrc/step16.hip make every launch
          of the 1024-stream cycle exactly 512 equal workgroups, without partial sums and a reduction launch.)
        * Latency mode (``split_k`` > 1, meant for a handful of streams): up to 4 * split_k ranges (<= 32), at least one
          8-channel chunk each -- a 9-tap chunk is 3.8 us of MFMAs for one workgroup, and with a handful of tiles the other
          250 CUs are idle anyway.  (Until round 4 the factor was also capped by 256 // tiles, i.e. by the slab size: one
          stream got 32 ranges at C = 256 and sixteen streams 18 -- different summation orders for the same stream.)"""
        base = self.THROUGHPUT_SPLIT_K if self.out_channels >= 256 else 1
        if self.split_k <= 1:
            return base
        return max(base, min(4 * self.split_k, 32, -(-self.out_channels // 8)))

    def _pick_gcn_ksplit(self, p: int) -> int:
        """Split-K factor of this block's graph conv (csk_gcn_stage_splitk_f32), latency mode only: with a handful of
        streams one workgroup per tile walks all 3 * C_in / 8 K-chunks alone (52 us at C_in = 256 -- 60 % of a frame's
        latency at one stream, profiles/r04_latency_1stream.md).  A function of (C_in, split_k) only (``p`` is not used)."""
        if self.split_k <= 1 or type(self.gcn) is not GraphConvolution or self.in_channels < 16:
            return 1
        return max(1, min(4 * self.split_k, 32, -(-self.in_channels // 8)))

    def clean_state(self):
        if self._state is not None:
            self._state.zero_()

    def engine_advance(self, r: int, n_frames: int, V: int, flush: bool = False):
        """Consume ``r`` (<= MAX_CYCLE) frames already stored in ``xin[(s .. s+r-1) % HIST]`` (channel-major).  Returns
        ``(first_out_slot, n_emit)`` for the emissions of these frames, or None.  ``flush`` pushes zero
        post-GCN frames instead (end padding)."""
        st, k = self._state, self.kernel_size
        HIST, YRING, OUT = st.xin.shape[0], st.y.shape[0], st.out.shape[0]      # ring depths of this block's slab slice
        if self.precision != "f32":
            raise NotImplementedError(
                "precision 'bf16x3' covers the clip kernels only: in step mode every ring slot feeds ONE tap per emission, so "
                "the split kernel would stage twice the bytes per MFMA of the clip form and is bound by staging, not by the "
                "matrix pipe (priced in DESIGN.md); step with the default precision")
        if not 1 <= r <= st.max_in:
            raise ValueError(f"engine_advance handles 1..{st.max_in} frames per call of this block, got {r}")
        s0, p = st.s, st.p
        if not flush and self._fusable(r, s0, V):
            return self._fused_advance(n_frames, V)
        f = 0
        while f < r:                               # per-frame graph conv, one launch per non-wrapping slot run
            s = s0 + f
            run = min(r - f, HIST - s % HIST, YRING - s % YRING)
            if flush:
                st.y[s % YRING: s % YRING + run].zero_()
            elif st.gcn_ksplit > 1:
                self.gcn.stage(st.xin[s % HIST], st.y[s % YRING], n_seg=run, frames=n_frames,
                               x_strides=(self.in_channels * p, p), y_strides=(self.out_channels * p, p),
                               ksplit=st.gcn_ksplit, partial=st.partial)
            elif hasattr(self.gcn, "stage"):
                self.gcn.stage(st.xin[s % HIST], st.y[s % YRING], n_seg=run, frames=n_frames,
                               x_strides=(self.in_channels * p, p), y_strides=(self.out_channels * p, p))
            else:
                self._foreign_gcn_stage(st, s, run, n_frames, V)
            f += run
        first = next((s for s in range(s0, s0 + r) if s >= self.delay and (s - self.delay) % self.stride == 0), None)
        st.s += r
        if first is None:
            return None
        n_emit = (s0 + r - 1 - first) // self.stride + 1
        ops = self._packed_ops(st.y.device)
        lag = (k - 1) // 2              # emission s pairs with input frame s - 4 (co.Delay / residual_shrink)
        mode = {"none": 0, "identity": 1, "conv": 2}[self.kind]
        slot0 = st.e % OUT
        # one launch; with split-K at most max_emit emissions per launch (the scratch holds that many partial sums) --
        # only the end-padding flush of a stack exceeds it (per-output summation order does not depend on the grouping)
        group = n_emit if (st.partial is None or st.ksplit <= 1) else min(n_emit, st.max_emit)     # only a split temporal conv is bound by the scratch
        for e0 in range(0, n_emit, group):
            ne, f0 = min(group, n_emit - e0), first + e0 * self.stride
            blocks.tcn_step_launch(
                native.ptr(st.y), YRING, f0 % YRING, self.stride, ne, native.ptr(ops["w"]),
                native.ptr(st.xin) if mode else None, HIST, (f0 - lag) % HIST, self.stride,
                native.ptr(ops["w_res"]), native.ptr(ops["bias"]), native.ptr(st.out), OUT, (slot0 + e0) % OUT,
                self.out_channels, self.out_channels, p, k, mode, self.in_channels if mode else 0, 1,
                st.ksplit, native.ptr(st.partial) if st.partial is not None else None, native.stream_of(st.y))
        st.e += n_emit
        return slot0, n_emit

    fuse_step = True    # one launch per block and stride cycle where csk_co_block_step_f32 applies (bit-identical)

    def _fusable(self, r: int, s0: int, V: int) -> bool:
        """csk_co_block_step_f32 (include/cskel.h): 64-row blocks, stride 1, a whole 4-frame cycle of emitting steps,
        native sparse graph conv, block residual none / identity, no split-K."""
        if not (self.fuse_step and r == 4 and self.stride == 1 and self.out_channels <= 64 and s0 >= self.delay
                and self.kind in ("none", "identity") and type(self.gcn) is GraphConvolution and self._state.ksplit == 1
                and self._state.gcn_ksplit == 1):
            return False
        g = self.gcn._packed_ops(self._state.y.device)
        cnt = g["ell_cnt_host"]
        return int(cnt[0]) <= 1 and int(cnt[1]) <= 1 and int(cnt[2]) <= 4 and ((64 + V - 2) // V + 1) * V <= 128

    def _fused_advance(self, n_skel: int, V: int):
        st = self._state
        HIST, YRING, OUT = st.xin.shape[0], st.y.shape[0], st.out.shape[0]
        g, t = self.gcn._packed_ops(st.y.device), self._packed_ops(st.y.device)
        s0, slot0 = st.s, st.e % OUT
        rc = native.lib().csk_co_block_step_f32(
            native.ptr(st.xin), HIST, s0 % HIST, self.in_channels, native.ptr(g["w"]), native.ptr(g["bias"]),
            native.ptr(g["ell_src"]), native.ptr(g["ell_val"]), native.ptr(g["ell_cnt_host"]), g["ell_w"], g["res_mode"],
            native.ptr(st.y), YRING, s0 % YRING, native.ptr(t["w"]), native.ptr(t["bias"]),
            {"none": 0, "identity": 1}[self.kind], (s0 - (self.kernel_size - 1) // 2) % HIST, native.ptr(st.out), OUT, slot0,
            self.out_channels, n_skel, V, st.p, native.stream_of(st.y))
        native.check(rc, "csk_co_block_step_f32")
        st.s += 4
        st.e += 4
        return slot0, 4

    def _foreign_gcn_stage(self, st, s: int, run: int, n_frames: int, V: int):
        """Graph-conv modules without a native ``stage`` (e.g. the S-TR spatial attention a sibling model passes as
        ``CoGraphConv``, models/base.py:390-400): applied per frame as ``module(x_t.unsqueeze(2)).squeeze(2)``
        (base.py:273-276) on (NM, C, 1, V) tensors converted from / to the channel-major ring slots."""
        q = n_frames * V
        HIST, YRING = st.xin.shape[0], st.y.shape[0]
        for j in range(run):
            xs, ys = st.xin[(s + j) % HIST], st.y[(s + j) % YRING]
            x_t = xs[:, :q].reshape(self.in_channels, n_frames, V).permute(1, 0, 2).unsqueeze(2).contiguous()
            y_t = self.gcn(x_t)
            if tuple(y_t.shape) != (n_frames, self.out_channels, 1, V) or y_t.dtype != torch.float32 or y_t.device != xs.device:
                raise RuntimeError(f"graph-conv module returned {tuple(y_t.shape)} {y_t.dtype} on {y_t.device}, expected "
                                   f"{(n_frames, self.out_channels, 1, V)} float32 on {xs.device}")
            ys[:, :q] = y_t.squeeze(2).permute(1, 0, 2).reshape(self.out_channels, q)

    def engine_step(self, n_frames: int, V: int, flush: bool = False) -> Optional[int]:
        """One frame; returns the output-ring slot of this step's emission or None."""
        res = self.engine_advance(1, n_frames, V, flush)
        return None if res is None else res[0]

    # ---- continual interface on (N, C, V) frames (module boundary: converts layouts) ------------
    def _ensure_state(self, n, v, device):
        p = _round4(n * v)
        if self._state is None or self._state.p != p or self._state.y.device != device or not self._state.owns_xin:
            self.bind_state(p, device)
        return self._state

    def forward_step(self, x_t, update_state=True):
        self._require_eval()
        native.require_device_f32(x_t, "CoSpatioTemporalBlock frame")
        n, c, v = x_t.shape
        if c != self.in_channels:
            raise RuntimeError(f"expected (N, {self.in_channels}, V) frame, got {tuple(x_t.shape)}")
        st = self._ensure_state(n, v, x_t.device)
        keep = (st.s, st.e)
        st.xin[st.s % st.xin.shape[0], :, : n * v] = x_t.permute(1, 0, 2).reshape(c, n * v)
        slot = self.engine_step(n, v)
        if not update_state:       # one step only touches ring slots that are older than every window: counters suffice
            st.s, st.e = keep
        if slot is None:
            return None
        return st.out[slot, :, : n * v].view(self.out_channels, n, v).permute(1, 0, 2).contiguous()

    def forward_steps(self, x, pad_end=False, update_state=True):
        n, c, t, v = x.shape
        if not update_state:                       # several frames overwrite live window slots: keep a copy
            st = self._ensure_state(n, v, x.device)
            keep = (st.y.clone(), st.out.clone(), st.xin.clone(), st.s, st.e)
            try:
                return self.forward_steps(x, pad_end, True)
            finally:
                st.y.copy_(keep[0]); st.out.copy_(keep[1]); st.xin.copy_(keep[2])
                st.s, st.e = keep[3], keep[4]
        outs = [o for o in (self.forward_step(x[:, :, i].contiguous()) for i in range(t)) if o is not None]
        if pad_end:
            st = self._state
            for _ in range(self.padding):
                slot = self.engine_step(n, v, flush=True)
                if slot is not None:
                    outs.append(st.out[slot, :, : n * v].view(self.out_channels, n, v).permute(1, 0, 2).contiguous())
        return torch.stack(outs, dim=2)


def co_geometry(c_in=3):
    """receptive_field / padding / stride of the ten-block stack (read from co.Sequential at base.py:86-97)."""
    r, p, s = 1, 0, 1
    for (_, _, st, _) in layer_table(c_in):
        r += 8 * s
        p += 4 * s
        s *= st
    return r, p, s


class CoStGcn(_Folded):
    """CoST-GCN: models/cost_gcn/cost_gcn.py:21-41 + CoModelBase (models/base.py:68-227) without the Ride shell.

    ``forward_step(x_t: (N, C, V, M))`` -> logits (N, classes) on the steps where the whole stack (10 blocks,
    total stride 4) and the temporal average pool emit, else None.  ``forward_steps(x: (N, C, T, V, M))``
    -> (N, classes, n_predictions).  ``forward(x)`` = clip mode of CoModelBase.forward (base.py:166-181).
    state_dict keys equal the reference's (``layers.layerK.0.1.gcn...``); a regular StGcn state_dict loads too
    (what ``map_state_dict`` does in the reference, base.py:200-224).
    """

    use_native_plan = True      # False: drive every launch from Python (same kernels, same results)

    def __init__(self, graph_A, input_shape=(3, 300, 25, 2), num_classes=60, pool_size=-1, pool_padding=-1,
                 CoGraphConv=CoGraphConvolution):
        super().__init__()
        (c_in, t, v, m) = input_shape
        self.input_shape, self.num_classes = tuple(input_shape), num_classes
        self.data_bn = nn.BatchNorm1d(m * c_in * v)
        self.layers = nn.ModuleDict(OrderedDict(
            (f"layer{i + 1}", CoSpatioTemporalBlock(ci, co, graph_A, stride=s, residual=r, padding="equal",
                                                    CoGraphConv=CoGraphConv))
            for i, (ci, co, s, r) in enumerate(layer_table(c_in))))
        self.fc = nn.Linear(256, num_classes)
        init_weights(self.data_bn, bs=1)
        init_weights(self.fc, bs=num_classes)
        self.receptive_field, self.padding, self.stride = co_geometry(c_in)
        self.delay = self.padding
        if pool_size == -1:                                                   # base.py:86-90
            pool_size = math.ceil((t - self.receptive_field + 2 * self.padding + 1) / self.stride)
        if pool_padding == -1:                                                # base.py:92-96
            pool_padding = pool_size - math.ceil((t - self.receptive_field + self.padding + 1) / self.stride)
        self.pool_size, self.pool_padding = pool_size, max(0, pool_padding)
        self._n = None
        self._flushed = False

    # ---- weights ---------------------------------------------------------------------------------
    def map_state_dict(self, state_dict, strict=True):
        """Regular-layout keys -> this module's (reference Co) layout (models/base.py:200-224).  A state dict
        that already holds every key of this module is returned unchanged, as in the reference."""
        own = nn.Module.state_dict(self).keys()
        if not (own - state_dict.keys()):
            return state_dict

        def short(k):
            return k.replace("0.1.", "").replace("0.0.residual", "residual")
        short2long = {short(k): k for k in own}
        return OrderedDict((short2long[k], v) for k, v in state_dict.items() if strict or k in short2long)

    def map_loaded_weights(self, file, loaded_state_dict):
        """Hook called by the checkpoint loader (models/base.py:226-227; weights.load_pretrained here)."""
        return self.map_state_dict(loaded_state_dict)

    def _fold(self):
        s, t = fold.fold_data_bn({k: v for k, v in nn.Module.state_dict(self).items() if k.startswith("data_bn.")})
        return dict(scale=s, shift=t)

    def _watched(self):
        return [self.data_bn]

    # ---- state slab --------------------------------------------------------------------------------
    def _bind(self, n, device):
        c_in, _, v, m = self.input_shape
        p = _round4(n * m * v)
        mc = self.max_cycle
        xin = torch.zeros((in_slots(mc), c_in, p), device=device, dtype=torch.float32)
        self._xin0, self._p, self._n = xin, p, n
        # frames one launch of block i can receive / emit: max_cycle input frames / cumulative temporal stride.  They size
        # the rings (y: 8 + max_in, output = next block's input history: 4 + its max_in) and the split-K scratch, which is
        # ONE buffer sized by its largest user (launches of a model are stream-ordered)
        recv, emits, cum = [], [], 1
        for i in range(10):
            recv.append(max(1, mc // cum))
            cum *= self.layers[f"layer{i + 1}"].stride
            emits.append(max(1, mc // cum))
        need = max(self.layers[f"layer{i + 1}"].scratch_floats(p, emits[i], recv[i]) for i in range(10))
        if need * 4 > self.LATENCY_SCRATCH_CAP_BYTES:
            raise RuntimeError(
                f"set_latency_mode({self.layers.layer1.split_k}) on a slab of {n} streams needs a {need * 4 / 1e9:.2f} GB split-K "
                f"scratch (cap {self.LATENCY_SCRATCH_CAP_BYTES / 1e9:.1f} GB): the latency mode splits every K loop into up to 32 "
                "channel ranges whatever the slab size -- it is meant for a handful of streams; use the default mode "
                "(set_latency_mode(0)) for slabs that fill the GPU")
        self._scratch = torch.empty((need,), device=device, dtype=torch.float32) if need else None
        for i in range(10):
            out_slots = in_slots(recv[i + 1]) if i < 9 else max(4, emits[i])
            st = self.layers[f"layer{i + 1}"].bind_state(p, device, xin, max_emit=emits[i], scratch=self._scratch,
                                                         max_in=recv[i], out_slots=out_slots)
            xin = st.out
        self._pool_ring = torch.zeros((self.pool_size, n, 256), device=device, dtype=torch.float32)
        self._pooled = torch.empty((n, 256), device=device, dtype=torch.float32)
        self._frames = self._feats = 0
        self._flushed = False
        self._build_plan(device)

    max_cycle = MAX_CYCLE   # frames ONE forward_cycle may carry: sizes the state rings (set_max_cycle)

    def set_max_cycle(self, frames: int = MAX_CYCLE):
        """Largest launch cycle (1..8 frames) the state slab is sized for.  A block's rings hold the 8-frame window of its
        temporal conv / the 4-frame residual lag PLUS the frames one launch brings (``max_cycle`` / cumulative stride), so the
        default of 8 pays for cycles the 4-frames-per-launch mode never issues: 5.75 GB at 1024 NTU streams against 4.65 GB with
        ``set_max_cycle(4)`` (SURVEY 8a's per-frame minimum: 3.25 GB).  Results do not depend on it (a frame lives in slot
        s % depth, the kernels take the depths as arguments).  Takes effect from a clean state (the slab is re-bound)."""
        if not isinstance(frames, int) or not 1 <= frames <= MAX_CYCLE:
            raise ValueError(f"max_cycle must be an integer in [1, {MAX_CYCLE}]")
        self.max_cycle = frames
        self._n = None

    LATENCY_SCRATCH_CAP_BYTES = 1 << 30   # split-K scratch above which binding a slab in latency mode is refused

    def set_latency_mode(self, split_k: int = 8):
        """Few-stream operation (one camera, a handful of streams): the TCN step of a block then holds one workgroup
        per tile that walks all 9*C/8 K-chunks alone (123 us at C = 256).  With ``split_k`` > 1 such launches cut the
        channel axis into up to ``split_k`` ranges computed by separate workgroups and summed in a fixed order
        (csk_tcn_step_f32 ``ksplit``); launches that fill the GPU anyway are left alone.  Results differ from the
        default by fp32 summation order only.  Takes effect from a clean state (the slab is re-bound).  (Replaying a
        frame's launches from hipGraphs was built and measured slower than eager launches on ROCm 7.2 -- 0.41 vs 0.36 ms per
        frame-step -- and removed: profiles/HISTORY.md.)  The split factor does not shrink with the slab (a stream's bits must
        not depend on its neighbours), so the scratch grows with it: binding a slab whose scratch would exceed
        LATENCY_SCRATCH_CAP_BYTES (1 GiB: about 300 NTU streams at split_k = 8) raises instead of silently running slower than the default mode."""
        for i in range(10):
            self.layers[f"layer{i + 1}"].split_k = int(split_k)
        self._n = None

    # ---- native executor ---------------------------------------------------------------------------
    def _mark_weights_dirty(self, *args, **kwargs):
        self.__dict__["_weights_dirty"] = True

    def _install_dirty_hooks(self):
        """load_state_dict on the model or ANY sub-module and .to() / .float() / ... (``_apply``) flag the plan's operands
        as stale immediately; see _weights_changed for everything else."""
        if self.__dict__.get("_dirty_hooks"):
            return
        for m in self.modules():
            m.register_load_state_dict_post_hook(lambda mod, keys, net=self: net._mark_weights_dirty())
        self.__dict__["_dirty_hooks"] = True

    def _apply(self, fn, *args, **kwargs):
        self._mark_weights_dirty()
        return super()._apply(fn, *args, **kwargs)

    def _weight_slots(self):
        """Flat snapshot of where every parameter / buffer / sub-module of the model LIVES (owning ``_parameters`` /
        ``_buffers`` / ``_modules`` dict + name) with the tensor's identity, storage pointer and version counter at the
        time the native plan's operands were folded.  Walking the module tree costs ~0.65 ms per call; re-reading these
        ~240 + ~140 dict slots costs ~0.06 ms, so _weights_changed can afford to be exact on every cycle."""
        tensors, modules = [], []
        for m in self.modules():
            for d in (m._parameters, m._buffers):
                for name, t in d.items():
                    tensors.append((d, name, t, None if t is None else t.data_ptr(), None if t is None else t._version))
            for name, child in m._modules.items():
                modules.append((m._modules, name, child, len(m._modules)))
        return tensors, modules

    def _weights_changed(self) -> bool:
        """Staleness check of the native plan's operands, once per cycle, EXACT and immediate for every way the weights
        can change: load_state_dict / .to() (dirty flag set by hooks), a replaced Parameter or buffer
        (``net.fc.weight = nn.Parameter(..)``: the slot holds another object), a swapped, added or removed sub-module
        (``net.layers.layer3.tcn.bn = ...``: the ``_modules`` slot holds another object / the dict changed size), an
        in-place edit (``p.add_(..)``: version counter) and ``p.data = ...`` (storage pointer)."""
        if self.__dict__.pop("_weights_dirty", False):
            return True
        tensors, modules = self._plan_keep[1]
        for d, name, t, ptr, ver in tensors:
            cur = d.get(name)
            if cur is not t or (t is not None and (cur._version != ver or cur.data_ptr() != ptr)):
                return True
        for d, name, child, size in modules:
            if d.get(name) is not child or len(d) != size:
                return True
        return False

    def refold(self):
        super().refold()
        self._mark_weights_dirty()

    def _layer_structs(self, device):
        """(ctypes array of csk_co_layer, objects to keep alive) from the blocks' packed operands and state."""
        arr, keep = (native.CoLayer * 10)(), []
        for i in range(10):
            blk = self.layers[f"layer{i + 1}"]
            g, t, st = blk.gcn._packed_ops(device), blk._packed_ops(device), blk._state
            keep += [g, t]
            L = arr[i]
            L.c_in, L.c_out, L.stride = blk.in_channels, blk.out_channels, blk.stride
            L.res_kind = {"none": 0, "identity": 1, "conv": 2}[blk.kind]
            L.gcn_res_mode, L.ell_w = g["res_mode"], g["ell_w"]
            for j in range(3):
                L.ell_cnt[j] = int(g["ell_cnt_host"][j])
            L.gcn_w, L.gcn_bias = g["w"].data_ptr(), g["bias"].data_ptr()
            L.ell_src = g["ell_src"].data_ptr()
            L.ell_val = g["ell_val"].data_ptr() if g["ell_val"] is not None else None
            L.tcn_w, L.tcn_bias = t["w"].data_ptr(), t["bias"].data_ptr()
            L.tcn_w_res = t["w_res"].data_ptr() if t["w_res"] is not None else None
            L.y_ring, L.out_ring = st.y.data_ptr(), st.out.data_ptr()
            L.tcn_ksplit = st.ksplit
            L.y_slots, L.out_slots = st.y.shape[0], st.out.shape[0]
            L.partial_emits = st.max_emit if (st.partial is not None and st.ksplit > 1) else 0
            L.gcn_ksplit, L.gcn_partial_frames = st.gcn_ksplit, (st.max_in if st.gcn_ksplit > 1 else 0)
            L.tcn_partial = st.partial.data_ptr() if st.partial is not None else None
            L.agcn_inter = L.agcn_adj_frames = 0
            if type(blk.gcn) is not GraphConvolution:      # adaptive graph conv: adjacency per skeleton frame (agcn.py)
                a = blk.gcn.plan_operands(device)
                adj = self.__dict__.get("_agcn_adj")
                need = self.max_cycle * self._n * self.input_shape[3] * 3 * self.input_shape[2] ** 2      # [cycle frames][skeletons][3][V][V]
                if adj is None or adj.numel() < need or adj.device != st.y.device:
                    adj = self.__dict__["_agcn_adj"] = torch.empty((need,), device=st.y.device, dtype=torch.float32)
                L.agcn_inter, L.agcn_adj_frames = a["inter"], self.max_cycle
                L.agcn_w_pairs, L.agcn_b_pairs, L.agcn_a_sum = a["w_pairs"].data_ptr(), a["b_pairs"].data_ptr(), a["a_sum"].data_ptr()
                L.agcn_adj = adj.data_ptr()
                L.ell_val = None
        ops = self._packed_ops(device)
        fcw, fcb = self.fc.weight.detach(), self.fc.bias.detach()
        keep += [ops, fcw, fcb]
        return arr, keep, ops, fcw, fcb

    def _build_plan(self, device):
        """csk_co_plan (include/cskel.h): one C call per cycle instead of ~25 (CoAGCN: ~45) ctypes calls.  Built for stacks of
        plain GraphConvolution blocks and of adaptive graph convs in the shapes the fused embedding + attention entry
        covers (``plan_operands``); other graph convs keep the Python engine below."""
        self._destroy_plan()
        if not self.use_native_plan:
            return
        for i in range(10):
            gcn = self.layers[f"layer{i + 1}"].gcn
            if type(gcn) is GraphConvolution:
                continue
            if getattr(gcn, "plan_operands", None) is None or gcn.plan_operands(device) is None:
                return
        c, _, v, m = self.input_shape
        arr, keep, ops, fcw, fcb = self._layer_structs(device)
        plan = native.lib().csk_co_plan_create(10, ctypes.byref(arr), native.ptr(self._xin0), self._xin0.shape[0], self._n, c, v, m, self._p,
                                               native.ptr(ops["scale"]), native.ptr(ops["shift"]), self.num_classes,
                                               native.ptr(fcw), native.ptr(fcb), self.pool_size, self.pool_padding,
                                               native.ptr(self._pool_ring), native.ptr(self._pooled))
        if not plan:
            raise RuntimeError("csk_co_plan_create: " + native.lib().csk_last_error().decode())
        self.__dict__["_plan"] = plan
        self._install_dirty_hooks()
        self.__dict__["_plan_keep"] = (keep, self._weight_slots())
        self.__dict__.pop("_weights_dirty", None)
        fuse = all(self.layers[f"layer{i + 1}"].fuse_step for i in range(10))
        native.check(native.lib().csk_co_plan_set_fusion(plan, int(fuse)), "csk_co_plan_set_fusion")

    def _refresh_plan_weights(self, device):
        """Weights were reloaded / edited in place: refold and hand the new operands to the plan; the
        continual state and its counters are untouched (same semantics as the reference, where weights and
        state buffers are independent)."""
        arr, keep, ops, fcw, fcb = self._layer_structs(device)
        rc = native.lib().csk_co_plan_update_weights(self._plan, 10, ctypes.byref(arr), native.ptr(ops["scale"]),
                                                     native.ptr(ops["shift"]), native.ptr(fcw), native.ptr(fcb))
        native.check(rc, "csk_co_plan_update_weights")
        self.__dict__["_plan_keep"] = (keep, self._weight_slots())

    def _destroy_plan(self):
        plan = self.__dict__.pop("_plan", None)
        if plan:
            native.lib().csk_co_plan_destroy(plan)
        self.__dict__.pop("_plan_keep", None)

    def __del__(self):
        try:
            self._destroy_plan()
        except Exception:
            pass

    def state_bytes(self):
        """Persistent continual state (input ring, per-block rings, pooling window)."""
        return sum(self.layers[f"layer{i + 1}"]._state.nbytes() for i in range(10)) + 4 * (
            self._xin0.numel() + self._pool_ring.numel())

    def scratch_bytes(self):
        """Transient scratch, not state: the split-K partial sums (shared by the blocks that split their K loop) and, for
        adaptive graph convs, the per-skeleton-frame adjacencies of a launch (shared by all blocks)."""
        adj = self.__dict__.get("_agcn_adj")
        return 4 * ((self._scratch.numel() if self._scratch is not None else 0) + (adj.numel() if adj is not None else 0))

    def clean_state(self):
        if self._n is not None:
            self._xin0.zero_()
            for i in range(10):
                self.layers[f"layer{i + 1}"].clean_state()
            self._pool_ring.zero_()
            self._frames = self._feats = 0
            self._flushed = False
            if self.__dict__.get("_plan"):
                native.lib().csk_co_plan_reset(self._plan)

    # ---- stepping ------------------------------------------------------------------------------------
    def _cycle(self, frames):
        """Advance by 1..MAX_CYCLE frames (list of (N, C, V, M) tensors): data_bn, ten blocks, head.
        Returns (slot, n_feat, logits): layer 10's emissions of this cycle (first output-ring slot, count;
        (None, 0) if none) and the list of predictions."""
        self._require_eval()
        frames = list(frames)
        if not 1 <= len(frames) <= self.max_cycle:
            raise ValueError(f"a cycle holds 1..{self.max_cycle} frames (set_max_cycle), got {len(frames)}")
        x0 = frames[0]
        for x_t in frames:
            native.require_device_f32(x_t, "CoStGcn frame")
            if x_t.shape != x0.shape or x_t.device != x0.device:
                raise RuntimeError("all frames of a cycle must have the same shape and device")
        n, c, v, m = x0.shape
        if (c, v, m) != (self.input_shape[0], self.input_shape[2], self.input_shape[3]):
            raise RuntimeError(f"frame shape {tuple(x0.shape)} does not match input_shape {self.input_shape}")
        if self._n != n or self._xin0.device != x0.device:           # clean_state_on_shape_change (base.py:161-164)
            self._bind(n, x0.device)
        if any(self.layers[f"layer{i + 1}"].precision != "f32" for i in range(10)):
            raise NotImplementedError("precision 'bf16x3' covers the clip kernels only (DESIGN.md section 4): step with the default "
                                      "precision -- set_precision(model, 'f32')")
        if self._flushed:
            raise RuntimeError("the state was flushed by forward_steps(pad_end=True): the end padding has consumed ring slots "
                               "and advanced the blocks past the input frame count; call clean_state() before stepping on")
        if self.__dict__.get("_plan"):
            return self._plan_cycle(frames)
        return self._python_cycle(frames)

    # ---- update_state=False (base.py:183-190 hand the flag through to co.Sequential) -------------------
    def _counters(self):
        snap = dict(frames=self._frames, feats=self._feats,
                    layers=[(b._state.s, b._state.e) for b in (self.layers[f"layer{i + 1}"] for i in range(10))])
        if self.__dict__.get("_plan"):
            buf = (ctypes.c_int64 * 22)()
            native.check(native.lib().csk_co_plan_counters(self._plan, buf, 22, 0), "csk_co_plan_counters")
            snap["plan"] = list(buf)
        return snap

    def _set_counters(self, snap):
        self._frames, self._feats = snap["frames"], snap["feats"]
        for i, (s_, e_) in enumerate(snap["layers"]):
            st = self.layers[f"layer{i + 1}"]._state
            st.s, st.e = s_, e_
        if "plan" in snap and self.__dict__.get("_plan"):
            buf = (ctypes.c_int64 * 22)(*snap["plan"])
            native.check(native.lib().csk_co_plan_counters(self._plan, buf, 22, 1), "csk_co_plan_counters")

    def _state_tensors(self):
        ts = [self._xin0, self._pool_ring, self._pooled]
        for i in range(10):
            st = self.layers[f"layer{i + 1}"]._state
            ts += [st.y, st.out]
        return ts

    def _ensure_bound(self, x_t):
        native.require_device_f32(x_t, "CoStGcn frame")
        if self._n != x_t.shape[0] or self._xin0.device != x_t.device:
            self._bind(x_t.shape[0], x_t.device)

    def _plan_cycle(self, frames):
        if self._weights_changed():
            self._refresh_plan_weights(frames[0].device)
        n = frames[0].shape[0]
        ptrs = (ctypes.c_void_p * len(frames))(*[x_t.data_ptr() for x_t in frames])
        logits = torch.empty((MAX_CYCLE, n, self.num_classes), device=frames[0].device, dtype=torch.float32)
        slot, nf, nl = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        rc = native.lib().csk_co_plan_cycle(self._plan, ptrs, len(frames), native.ptr(logits), ctypes.byref(slot),
                                            ctypes.byref(nf), ctypes.byref(nl), native.stream_of(frames[0]))
        native.check(rc, "csk_co_plan_cycle")      # on failure the plan has put its counters back (executor.hip)
        self._frames += len(frames)
        self._feats += nf.value
        if nf.value == 0:
            return None, 0, []
        return slot.value, nf.value, [logits[j] for j in range(nl.value)]

    def _python_cycle(self, frames):
        """Same protocol driven from Python (any graph-conv module with a ``stage`` method)."""
        n, c, v, m = frames[0].shape
        ops = self._packed_ops(frames[0].device)
        # reshape1 + data_bn + reshape2 (base.py:73-82) of the cycle's frames straight into the channel-major input ring
        depth = self._xin0.shape[0]
        srcs = (ctypes.c_void_p * len(frames))(*[x_t.data_ptr() for x_t in frames])
        dsts = (ctypes.c_void_p * len(frames))(*[self._xin0[(self._frames + f) % depth].data_ptr() for f in range(len(frames))])
        rc = native.lib().csk_input_norm_frames_f32(srcs, dsts, len(frames), native.ptr(ops["scale"]), native.ptr(ops["shift"]),
                                                    n, c, v, m, self._p, native.stream_of(frames[0]))
        native.check(rc, "csk_input_norm_frames_f32")
        self._frames += len(frames)
        r, res = len(frames), None
        for i in range(10):
            res = self.layers[f"layer{i + 1}"].engine_advance(r, n * m, v)
            if res is None:
                return None, 0, []
            r = res[1]
        outs, depth = [], self.layers["layer10"]._state.out.shape[0]
        for j in range(res[1]):
            o = self._head_step((res[0] + j) % depth, n)
            if o is not None:
                outs.append(o)
        return res[0], res[1], outs

    def _head_step(self, slot, n):
        """spatial_pool -> co.AvgPool1d window -> co.Linear (base.py:84-101)."""
        _, _, v, m = self.input_shape
        st10 = self.layers["layer10"]._state
        lib = native.lib()
        head = self._feats % self.pool_size
        stream = native.stream_of(st10.out)
        self._feats += 1
        emit = self._feats >= self.pool_size - self.pool_padding
        count = min(self._feats, self.pool_size)
        logits = torch.empty((n, self.num_classes), device=st10.out.device, dtype=torch.float32) if emit else None
        # slot None: end padding of the pooling window (a zero feature enters it)
        native.check(lib.csk_co_head_step_f32(
            native.ptr(st10.out[slot]) if slot is not None else None, native.ptr(self._pool_ring), native.ptr(self._pooled),
            native.ptr(self.fc.weight.detach()), native.ptr(self.fc.bias.detach()), native.ptr(logits), n, 256, m * v, self._p,
            self.pool_size, head, count, int(emit), self.num_classes, stream), "csk_co_head_step_f32")
        return logits

    def features_step(self, x_t):
        """(N, C, V, M) frame -> slot of layer 10's output ring holding this step's emission, or None
        (the head advances as well, exactly as in ``forward_step``)."""
        return self._cycle([x_t])[0]

    def forward_step(self, x_t, update_state=True):
        """CoModelBase.forward_step (base.py:183-185): logits (N, classes) on predicting steps, else None.
        ``update_state=False`` computes the step without advancing: a single step only overwrites ring slots whose
        content has left every window (and the oldest entry of the pooling window, which the next real step replaces
        as well), so restoring the counters restores the state."""
        if update_state:
            outs = self._cycle([x_t])[2]
            return outs[-1] if outs else None
        self._ensure_bound(x_t)
        snap = self._counters()
        try:
            outs = self._cycle([x_t])[2]
        finally:
            self._set_counters(snap)
        return outs[-1] if outs else None

    def forward_cycle(self, frames):
        """Up to MAX_CYCLE (8) consecutive frames in one go (list of (N, C, V, M) tensors): same results as calling
        ``forward_step`` on each, with len(frames)x fewer and larger launches (4 = one stride cycle of the stack).  Returns the list of logits emitted."""
        return self._cycle(frames)[2]

    def forward_steps(self, x, pad_end=False, update_state=True):
        """(N, C, T, V, M) -> (N, classes, n_predictions) (empty last dim if nothing was emitted).  ``pad_end`` is handed
        through to every module as in the reference (base.py:187-190): each block's temporal conv is flushed with its
        ``padding`` zero frames, first block first, so that the stack emits what the clip stack computes for the same
        frames, and the temporal average pool is flushed with ``pool_padding`` zero features."""
        if not update_state:                       # several frames overwrite live window slots: keep a copy of the slab
            self._ensure_bound(x[:, :, 0].contiguous())
            snap, keep, flushed = self._counters(), [t.clone() for t in self._state_tensors()], self._flushed
            try:
                return self.forward_steps(x, pad_end, True)
            finally:
                for t, k in zip(self._state_tensors(), keep):
                    t.copy_(k)
                self._set_counters(snap)
                self._flushed = flushed
        outs = []
        for t in range(x.shape[2]):
            o = self.forward_step(x[:, :, t].contiguous())
            if o is not None:
                outs.append(o)
        if pad_end and x.shape[2] > 0:
            outs += self._flush()
            self._flushed = True                   # stepping on needs clean_state() (see _flush)
        if not outs:
            return torch.empty((x.shape[0], self.num_classes, 0), device=x.device)
        return torch.stack(outs, dim=2)

    def _flush(self):
        """End padding of the whole model (``pad_end=True``): returns the predictions it releases.  Runs on the Python
        engine; a native plan's counters are read before and written back afterwards.  The flush ends the sequence: it
        zeroes y-ring slots and advances the per-block counters by their padding while the input frame count stays,
        so the rings no longer line up with ``frames % depth`` -- the model is marked flushed and the next step raises
        until ``clean_state()`` (continual-inference's own end padding does not save state either; a caller that wants
        to go on uses ``update_state=False``, which runs the flush on a snapshot)."""
        plan = self.__dict__.get("_plan")
        if plan:
            buf = (ctypes.c_int64 * 22)()
            native.check(native.lib().csk_co_plan_counters(plan, buf, 22, 0), "csk_co_plan_counters")
            self._frames, self._feats = int(buf[0]), int(buf[1])
            for i in range(10):
                st = self.layers[f"layer{i + 1}"]._state
                st.s, st.e = int(buf[2 + 2 * i]), int(buf[3 + 2 * i])
        n = self._n
        _, _, v, m = self.input_shape
        outs, depth = [], self.layers["layer10"]._state.out.shape[0]
        for i in range(10):
            blk = self.layers[f"layer{i + 1}"]
            left = blk.padding
            while left:                                # at most max_in frames per launch (ring depths); order is unchanged
                r = min(left, blk._state.max_in)
                left -= r
                res = blk.engine_advance(r, n * m, v, flush=True)
                for j in range(i + 1, 10):             # what block i released travels down the rest of the stack
                    if res is None:
                        break
                    res = self.layers[f"layer{j + 1}"].engine_advance(res[1], n * m, v)
                if res is not None:
                    for jj in range(res[1]):
                        o = self._head_step((res[0] + jj) % depth, n)
                        if o is not None:
                            outs.append(o)
        for _ in range(self.pool_padding):         # co.AvgPool1d end padding: zero features enter the window
            o = self._head_step(None, n)
            if o is not None:
                outs.append(o)
        if plan:
            self._set_counters(self._counters_py())
        return outs

    def _counters_py(self):
        snap = dict(frames=self._frames, feats=self._feats,
                    layers=[(b._state.s, b._state.e) for b in (self.layers[f"layer{i + 1}"] for i in range(10))])
        flat = [self._frames, self._feats]
        for s_, e_ in snap["layers"]:
            flat += [s_, e_]
        snap["plan"] = flat
        return snap

    def forward(self, x, forward_mode="clip"):
        """CoModelBase.forward (base.py:166-181).  'clip': whole-clip computation with the continual head
        (zero-padded temporal average pool, first window); 'frame': stepping with a fresh state."""
        self._require_eval()
        if forward_mode == "frame":
            self.clean_state()
            ret = self.forward_steps(x)
            return ret[:, :, 0]
        n, c, t, v, m = x.shape
        h = self._clip_features(x)                                        # (N*M, 256, T', V)
        tp = h.shape[2]
        # spatial_pool per frame, then AvgPool1d(pool_size, stride 1, padding) output index 0:
        # frames [0, pool_size - pool_padding) summed, divided by pool_size (zeros included)
        take = min(tp, self.pool_size - self.pool_padding)
        feat = torch.empty((n, 256), device=x.device, dtype=torch.float32)
        hs = h[:, :, :take].contiguous()
        rc = native.lib().csk_pool_scaled_f32(native.ptr(hs), native.ptr(feat), n, m, 256, take * v,
                                              take / self.pool_size, native.stream_of(x))
        native.check(rc, "csk_pool_scaled_f32")
        logits = torch.empty((n, self.num_classes), device=x.device, dtype=torch.float32)
        native.check(native.lib().csk_fc_f32(native.ptr(feat), native.ptr(self.fc.weight.detach()),
                                             native.ptr(self.fc.bias.detach()), native.ptr(logits), n, 256,
                                             self.num_classes, native.stream_of(x)), "csk_fc_f32")
        return logits

    def _clip_features(self, x):
        native.require_device_f32(x, "CoStGcn input")
        n, c, t, v, m = x.shape
        ops = self._packed_ops(x.device)
        h = torch.empty((n * m, c, t, v), device=x.device, dtype=torch.float32)
        rc = native.lib().csk_input_norm_f32(native.ptr(x), native.ptr(ops["scale"]), native.ptr(ops["shift"]),
                                             native.ptr(h), n, c, t, v, m, c * t * v, t * v, native.stream_of(x))
        native.check(rc, "csk_input_norm_f32")
        for i in range(10):
            h = SpatioTemporalBlock.forward(self.layers[f"layer{i + 1}"], h)
        return h

    def warm_up(self, n, device, frames=None):
        """Feed ``receptive_field - padding - 1`` random frames (models/base.py:144-159) so that the next
        frame produces layer-10 output."""
        self.clean_state()
        c, _, v, m = self.input_shape
        frames = self.receptive_field - self.padding - 1 if frames is None else frames
        for _ in range(frames):
            self.forward_step(torch.randn((n, c, v, m), device=device))
