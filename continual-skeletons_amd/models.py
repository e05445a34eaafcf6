"""Model drivers around the block library: the callers on either side of the hot path.

``StGcn`` is the counterpart of ``models/st_gcn/st_gcn.py:20-65`` minus the Ride/Lightning shell
(CLI, training, datasets are out of scope): same layer table, same attribute names
(``data_bn``, ``layers.layerK``, ``fc``) and therefore the same ``state_dict`` keys, so published
ST-GCN checkpoints load unchanged.  ``forward`` = input-norm kernel -> 10 fused blocks -> pool+fc kernels.
"""
import torch
import torch.nn as nn

from . import fold, native
from .blocks import SpatioTemporalBlock, _Folded, init_weights


def layer_table(c_in):
    """(in, out, stride, residual) of the ten blocks (models/st_gcn/st_gcn.py:30-39)."""
    return [
        (c_in, 64, 1, False), (64, 64, 1, True), (64, 64, 1, True), (64, 64, 1, True),
        (64, 128, 2, True), (128, 128, 1, True), (128, 128, 1, True),
        (128, 256, 2, True), (256, 256, 1, True), (256, 256, 1, True),
    ]


class StGcn(_Folded):
    def __init__(self, graph_A, input_shape=(3, 300, 25, 2), num_classes=60, GraphConv=None):
        """graph_A: (3, V, V) adjacency; input_shape = (C, T, V, M) as datasets/datasets.py:128-134."""
        super().__init__()
        (num_channels, num_frames, num_vertices, num_skeletons) = input_shape
        self.input_shape = tuple(input_shape)
        self.num_classes = num_classes
        kw = {} if GraphConv is None else {"GraphConv": GraphConv}
        self.data_bn = nn.BatchNorm1d(num_skeletons * num_channels * num_vertices)
        self.layers = nn.ModuleDict({
            f"layer{i + 1}": SpatioTemporalBlock(ci, co, graph_A, stride=s, residual=r, **kw)
            for i, (ci, co, s, r) in enumerate(layer_table(num_channels))
        })
        self.fc = nn.Linear(256, num_classes)
        init_weights(self.data_bn, bs=1)
        init_weights(self.fc, bs=num_classes)

    def _fold(self):
        s, t = fold.fold_data_bn({k: v for k, v in self.state_dict().items() if k.startswith("data_bn.")})
        return dict(scale=s, shift=t)

    def _watched(self):       # only the driver's own folded tensors; blocks keep their own caches
        return [self.data_bn]

    def input_norm(self, x):
        """(N, C, T, V, M) -> (N*M, C, T, V): permute + data_bn (models/st_gcn/st_gcn.py:49-57)."""
        native.require_device_f32(x, "StGcn input")
        n, c, t, v, m = x.shape
        ops = self._packed_ops(x.device)
        if ops["scale"].numel() != m * v * c:
            raise RuntimeError(f"input (C,V,M)=({c},{v},{m}) does not match data_bn with {ops['scale'].numel()} channels")
        h = torch.empty((n * m, c, t, v), device=x.device, dtype=torch.float32)
        rc = native.lib().csk_input_norm_f32(native.ptr(x), native.ptr(ops["scale"]), native.ptr(ops["shift"]),
                                             native.ptr(h), n, c, t, v, m, c * t * v, t * v, native.stream_of(x))
        native.check(rc, "csk_input_norm_f32")
        return h

    def features(self, x):
        h = self.input_norm(x)
        for i in range(len(self.layers)):
            h = self.layers[f"layer{i + 1}"](h)
        return h

    def head(self, h, n, m):
        """mean over (T, V), mean over M, fc (models/st_gcn/st_gcn.py:60-64)."""
        nm, c, t, v = h.shape
        feat = torch.empty((n, c), device=h.device, dtype=torch.float32)
        logits = torch.empty((n, self.num_classes), device=h.device, dtype=torch.float32)
        rc = native.lib().csk_pool_fc_f32(native.ptr(h), native.ptr(self.fc.weight.detach()), native.ptr(self.fc.bias.detach()),
                                          native.ptr(feat), native.ptr(logits), n, m, c, t * v, self.num_classes,
                                          native.stream_of(h))
        native.check(rc, "csk_pool_fc_f32")
        return logits

    def forward(self, x):
        self._require_eval()
        n, c, t, v, m = x.shape
        return self.head(self.features(x), n, m)

    def set_latency_mode(self, split_k: int = 4, gcn_split_k: int = None, max_sequences: int = None):
        """Small-batch clip inference: split the K loops of every block over workgroups for forwards of at most
        ``max_sequences`` sequences (N * M; default 6), the default kernels above that (blocks.set_clip_latency_mode)."""
        from .blocks import CLIP_SPLIT_MAX_SEQUENCES, set_clip_latency_mode
        return set_clip_latency_mode(self, split_k, gcn_split_k, CLIP_SPLIT_MAX_SEQUENCES if max_sequences is None else max_sequences)
