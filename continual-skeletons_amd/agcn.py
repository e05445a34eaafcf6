"""A-GCN adaptive graph convolution and the (Co)AGCN model drivers (BASELINE.json configs[3]).

Counterpart of ``AdaptiveGraphConvolution`` (models/a_gcn/a_gcn.py:12-69), ``AGcn`` (a_gcn.py:72-145) and
``CoAGcn`` (models/coa_gcn/coa_gcn.py).  The graph conv aggregates with a PER-SAMPLE dense adjacency
    adj_i = softmax_{dim=-2}( a_i(x)^T b_i(x) / (inter*T) ) + (A + graph_attn)_i        (a_gcn.py:50-63)
Three launches: (1) the six 1x1 embedding convs as ONE fused 1x1 conv (tcn stage / step kernel), (2)
``csk_agcn_attention_f32`` -> dense column-wise adjacency values per sample, (3) the general GCN stage kernel
with ``adj_seg_stride`` (aggregation on VALU from LDS-staged tables, channel mixing on MFMA).
Clip mode: one V x V matrix per sample over all T; continual mode (CoAGCN): T = 1, i.e. per-frame attention --
clip != step by design (a_gcn.py:60-61, coa_gcn.py:31).
"""
import torch
import torch.nn as nn

from . import blocks, fold, native
from .blocks import GraphConvolution, _check_input, init_weights
from .continual import CoStGcn
from .models import StGcn


class AdaptiveGraphConvolution(GraphConvolution):
    def __init__(self, in_channels, out_channels, A, bn_momentum=0.1, coff_embedding=4):
        super().__init__(in_channels, out_channels, A, bn_momentum)
        self.inter_c = out_channels // coff_embedding
        self.a_conv = nn.ModuleList(nn.Conv2d(in_channels, self.inter_c, 1) for _ in range(self.num_subset))
        self.b_conv = nn.ModuleList(nn.Conv2d(in_channels, self.inter_c, 1) for _ in range(self.num_subset))
        for m in list(self.a_conv) + list(self.b_conv):
            init_weights(m)
        self.soft = nn.Softmax(-2)

    def _fold(self):
        sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
        f = fold.fold_graph_conv(sd)                      # weights / bias as for GraphConvolution ...
        v = f["V"]
        # ... but the adjacency is dense and per sample: index pattern 0..V-1 for every column
        f["ell_src"] = torch.arange(v, dtype=torch.int32).repeat(3, v, 1).contiguous()
        f["ell_cnt_host"] = torch.tensor([v, v, v], dtype=torch.int32)
        f["ell_w"] = v
        f["ell_val"] = None
        f["a_sum"] = (sd["A"].double() + sd["graph_attn"].double()).float().contiguous()      # a_gcn.py:50
        inter, ci = self.inter_c, self.in_channels
        we = torch.cat([sd[f"a_conv.{i}.weight"] for i in range(3)] + [sd[f"b_conv.{i}.weight"] for i in range(3)], 0)
        be = torch.cat([sd[f"a_conv.{i}.bias"] for i in range(3)] + [sd[f"b_conv.{i}.bias"] for i in range(3)], 0)
        f["w_embed"] = fold.pack_conv_weight(we.view(6 * inter, ci, 1, 1), torch.ones(6 * inter, dtype=torch.float64))
        f["b_embed"] = fold.pad_vec(be.double())
        # pair-major row order (a_i rows, then b_i rows, i = 0, 1, 2) for the fused step kernel
        order = [r for i in range(3) for r in list(range(i * inter, (i + 1) * inter)) + list(range((3 + i) * inter, (4 + i) * inter))]
        f["w_embed_pairs"] = fold.pack_conv_weight(we[order].view(6 * inter, ci, 1, 1), torch.ones(6 * inter, dtype=torch.float64))
        f["b_embed_pairs"] = fold.pad_vec(be[order].double())
        return f

    def _attention(self, E, ops, n_seg, T, V, e_seg_stride, e_chan_stride, seg_per_group=None, e_group_stride=0):
        adj = torch.empty((n_seg, 3, V, V), device=E.device, dtype=torch.float32)
        scratch = torch.empty((n_seg, 3, 4, V, V), device=E.device, dtype=torch.float32) if T > 1 else None
        rc = native.lib().csk_agcn_attention_f32(native.ptr(E), native.ptr(ops["a_sum"]), native.ptr(adj), native.ptr(scratch), n_seg,
                                                 self.inter_c, T, V, e_seg_stride, e_chan_stride,
                                                 seg_per_group or n_seg, e_group_stride, native.stream_of(E))
        native.check(rc, "csk_agcn_attention_f32")
        return adj

    def forward(self, x):
        self._require_eval()
        _check_input(x, self.in_channels, "AdaptiveGraphConvolution input")
        ops = self._packed_ops(x.device)
        n, c, t, v = x.shape
        e_ch = 6 * self.inter_c
        y = torch.empty((n, self.out_channels, t, v), device=x.device, dtype=torch.float32)
        # (V = 25: the fused clip form measured SLOWER than GEMM + logits kernels -- 28.0 vs 24.2 ms per NTU batch-64 forward: five
        # frames per tile, joints staged one by one -- so it is used at V = 18 only; the step form gains 4 % at V = 25 too)
        if v == 18 and self._fused_ok(v, x.data_ptr(), c * t * v, t * v):
            # embedding convs + partial logits in one launch, softmax in a second (csk_agcn_embed_attention_f32, per-segment form)
            adj = torch.empty((n, 3, v, v), device=x.device, dtype=torch.float32)
            ft = 128 // v                              # frames per tile of the kernel
            scratch = torch.empty((n, 3, (t + ft - 1) // ft, v, v), device=x.device, dtype=torch.float32)
            rc = native.lib().csk_agcn_embed_attention_f32(
                native.ptr(x), native.ptr(ops["w_embed_pairs"]), native.ptr(ops["b_embed_pairs"]), native.ptr(ops["a_sum"]),
                native.ptr(adj), native.ptr(scratch), n, c, self.inter_c, t, v, 0, c * t * v, t * v, native.stream_of(x))
            native.check(rc, "csk_agcn_embed_attention_f32")
            blocks.gcn_stage(x, y, dict(ops, ell_val=adj), n_seg=n, frames=t, x_strides=(c * t * v, t * v),
                             y_strides=(self.out_channels * t * v, t * v), adj_seg_stride=3 * v * v)
            return y
        # (N, 6*inter, T, V), + 4 floats of slack behind it (csk_agcn_attention_f32 reads whole 16-byte vectors)
        E = torch.empty((n * e_ch * t * v + 4,), device=x.device, dtype=torch.float32)[: n * e_ch * t * v].view(n, e_ch, t, v)
        rc = native.lib().csk_conv1x1_f32(native.ptr(x), native.ptr(E), native.ptr(ops["w_embed"]), native.ptr(ops["b_embed"]), n, c,
                                          e_ch, t, v, c * t * v, t * v, e_ch * t * v, t * v, native.stream_of(x))
        native.check(rc, "csk_conv1x1_f32")
        adj = self._attention(E, ops, n, t, v, e_ch * t * v, t * v)
        o = dict(ops, ell_val=adj)
        blocks.gcn_stage(x, y, o, n_seg=n, frames=t, x_strides=(c * t * v, t * v),
                         y_strides=(self.out_channels * t * v, t * v), adj_seg_stride=3 * v * v)
        return y

    fuse_embed_attention = True     # False: two launches (csk_conv1x1_f32 + csk_agcn_attention_f32) -- for A/B measurements

    def _fused_ok(self, v, data_ptr, seg_stride, chan_stride):
        """Shapes csk_agcn_embed_attention_f32 is built for (even V: 8-byte aligned activation rows)."""
        if not self.fuse_embed_attention or v not in (18, 25) or self.inter_c not in (16, 32, 64):
            return False
        return v % 2 == 1 or (data_ptr % 8 == 0 and seg_stride % 2 == 0 and chan_stride % 2 == 0)

    def plan_operands(self, device):
        """Operands of the native step executor (csk_co_layer.agcn_*: the per-frame form of csk_agcn_embed_attention_f32), or
        None where that entry is not built (V not in {18, 25}, inter not in {16, 32, 64}): such stacks keep the Python engine."""
        ops = self._packed_ops(device)
        if not self.fuse_embed_attention or ops["V"] not in (18, 25) or self.inter_c not in (16, 32, 64):
            return None
        return dict(inter=self.inter_c, w_pairs=ops["w_embed_pairs"], b_pairs=ops["b_embed_pairs"], a_sum=ops["a_sum"])

    def stage(self, x, y, n_seg, frames, x_strides, y_strides):
        """Continual use on channel-major frames (C, P): every skeleton is its own 'sample' with T = 1, so the
        attention is computed per skeleton and frame (coa_gcn.py: forward_stepping of the module) and the graph
        conv runs with a per-frame adjacency.  The n_seg consecutive ring slots of a launch cycle go through three
        launches together: embedding conv (one emission per slot), attention, graph conv (one segment per slot)."""
        ops = self._packed_ops(x.device)
        v, p = ops["V"], x_strides[1]
        e_ch = 6 * self.inter_c
        if self._fused_ok(v, x.data_ptr(), x_strides[0], p):
            # embedding convs + attention in one launch (csk_agcn_embed_attention_f32, per-frame form), then the graph conv
            adj = torch.empty((n_seg * frames, 3, v, v), device=x.device, dtype=torch.float32)
            rc = native.lib().csk_agcn_embed_attention_f32(
                native.ptr(x), native.ptr(ops["w_embed_pairs"]), native.ptr(ops["b_embed_pairs"]), native.ptr(ops["a_sum"]),
                native.ptr(adj), None, n_seg, self.in_channels, self.inter_c, frames, v, 1, x_strides[0], p, native.stream_of(x))
            native.check(rc, "csk_agcn_embed_attention_f32")
            blocks.gcn_stage(x, y, dict(ops, ell_val=adj), n_seg=n_seg, frames=frames, x_strides=x_strides, y_strides=y_strides,
                             adj_seg_stride=3 * v * v, adj_per_frame=1)
            return
        E = torch.empty((n_seg, e_ch, p), device=x.device, dtype=torch.float32)
        # the fused 1x1 embedding conv on the channel-major slots: segment = ring slot, frames = skeletons
        rc = native.lib().csk_conv1x1_f32(native.ptr(x), native.ptr(E), native.ptr(ops["w_embed"]), native.ptr(ops["b_embed"]), n_seg,
                                          self.in_channels, e_ch, frames, v, x_strides[0], p, e_ch * p, p, native.stream_of(x))
        native.check(rc, "csk_conv1x1_f32")
        adj = self._attention(E, ops, n_seg * frames, 1, v, v, p, seg_per_group=frames, e_group_stride=e_ch * p)
        o = dict(ops, ell_val=adj)
        # segment = one channel-major frame, adjacency per "frame" of it (= skeleton): index seg * frames + skeleton
        blocks.gcn_stage(x, y, o, n_seg=n_seg, frames=frames, x_strides=x_strides, y_strides=y_strides,
                         adj_seg_stride=3 * v * v, adj_per_frame=1)


def CoAdaptiveGraphConvolution(in_channels, out_channels, A, bn_momentum=0.1):
    """models/coa_gcn/coa_gcn.py:11-14"""
    return AdaptiveGraphConvolution(in_channels, out_channels, A, bn_momentum)


class AGcn(StGcn):
    """models/a_gcn/a_gcn.py:72-145: the ST-GCN layer table with AdaptiveGraphConvolution."""

    def __init__(self, graph_A, input_shape=(3, 300, 25, 2), num_classes=60):
        super().__init__(graph_A, input_shape, num_classes, GraphConv=AdaptiveGraphConvolution)


class CoAGcn(CoStGcn):
    """models/coa_gcn/coa_gcn.py: CoST-GCN stack with the adaptive graph conv applied per frame."""

    def __init__(self, graph_A, input_shape=(3, 300, 25, 2), num_classes=60, pool_size=-1, pool_padding=-1):
        super().__init__(graph_A, input_shape, num_classes, pool_size, pool_padding, CoGraphConv=CoAdaptiveGraphConvolution)
