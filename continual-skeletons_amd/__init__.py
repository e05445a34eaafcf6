"""MI355X-native ST-GCN / CoST-GCN forward path.

Drop-in for the block library of LukasHedegaard/continual-skeletons ``models/base.py`` (reference
names exported unchanged; the north-star aliases ``SpatialGraphConv`` / ``StGcnBlock`` /
``CoStGcnBlock`` point at them).  All arithmetic runs in hand-written HIP kernels for gfx950 behind
the C ABI of ``include/cskel.h``; there is no CPU or PyTorch-op fallback -- a missing library or a
CPU tensor raises.

The directory name carries a hyphen, so it is imported through ``_bootstrap.load()`` (repo root),
which registers it as the module ``continual_skeletons_amd``.
"""
from .graph import Graph, kinetics_graph, ntu_graph  # noqa: F401
from .blocks import (  # noqa: F401
    GraphConvolution,
    SpatioTemporalBlock,
    TemporalConvolution,
    init_weights,
    set_clip_latency_mode,
    set_precision,
    unity,
    zero,
)
from .models import StGcn  # noqa: F401
from .continual import (  # noqa: F401
    CoGraphConvolution,
    CoSpatioTemporalBlock,
    CoStGcn,
    CoTemporalConvolution,
)
from .agcn import AdaptiveGraphConvolution, AGcn, CoAdaptiveGraphConvolution, CoAGcn  # noqa: F401
from . import fusion, native, weights  # noqa: F401
from .weights import load_pretrained  # noqa: F401

# names used by BASELINE.json:north_star
SpatialGraphConv = GraphConvolution
StGcnBlock = SpatioTemporalBlock
CoStGcnBlock = CoSpatioTemporalBlock

__all__ = [
    "Graph", "ntu_graph", "kinetics_graph", "GraphConvolution", "TemporalConvolution",
    "SpatioTemporalBlock", "SpatialGraphConv", "StGcnBlock", "CoStGcnBlock", "StGcn", "CoStGcn",
    "CoGraphConvolution", "CoTemporalConvolution", "CoSpatioTemporalBlock",
    "AdaptiveGraphConvolution", "CoAdaptiveGraphConvolution", "AGcn", "CoAGcn", "init_weights", "zero", "unity",
    "native", "fusion", "set_precision", "set_clip_latency_mode",
]
