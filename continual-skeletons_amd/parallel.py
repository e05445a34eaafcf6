"""Batch-axis sharding over the GPUs of one node (BASELINE.json configs[4]).

Every clip / stream is independent through all ten blocks (eval-mode BN is a per-channel affine, no
cross-sample op), so the N axis is partitioned into contiguous slices, weights are replicated
(12.6 MB) and continual state stays with its streams.  The one exchange step is an all-gather of the
logits ``(N/world, classes)`` -- RCCL over xGMI when the process group is "nccl"; 240 KiB per rank at
1024 clips/GPU, i.e. latency-bound, so it is issued as ONE collective (torch.distributed orders it after the
logits on the current stream and runs it on the process group's own stream; the next reader of the result waits on it).  The two skeletons (M) of a clip never split: sharding is over whole clips.
"""
import ctypes
import warnings

import torch
import torch.distributed as dist

from . import native


def shard_bounds(n_total: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of the clip axis owned by ``rank`` (sizes differ by at most one)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_logits(local: torch.Tensor, group=None) -> torch.Tensor:
    """(n_local, classes) on every rank -> (sum n_local, classes) on every rank, rank-major order.
    Requires equal n_local on all ranks (use ``all_gather_ragged`` otherwise)."""
    world = dist.get_world_size(group)
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out


def all_gather_ragged(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Same for uneven shards produced by ``shard_bounds``: pad to the largest shard, gather, trim."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0] for r in range(world)]
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} rows, expected {sizes[rank]}")
    width = max(sizes)
    padded = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    gathered = all_gather_logits(padded, group).view(world, width, *local.shape[1:])
    return torch.cat([gathered[r, : sizes[r]] for r in range(world)], dim=0)


def streams_overlap(a, b, spin_us: int = 200) -> bool:
    """True if HIP streams ``a`` and ``b`` run concurrently (they sit on different hardware queues)."""
    ratio = ctypes.c_float(0.0)
    native.check(native.lib().csk_stream_overlap_probe(a.cuda_stream, b.cuda_stream, spin_us, ctypes.byref(ratio)),
                 "csk_stream_overlap_probe")
    return ratio.value < 1.5


def concurrent_streams(n: int, device, candidates: int = 16):
    """``n`` HIP streams that run concurrently with each other and with the current stream.

    HIP multiplexes streams onto 4 hardware queues (GPU_MAX_HW_QUEUES) in creation order, and PyTorch hands out
    its pool streams round-robin, so which shards would share a queue -- and silently serialise -- depends on how
    many streams the process (RCCL included) created before.  Candidates are therefore probed
    (csk_stream_overlap_probe) and only mutually concurrent ones are kept.  If the hardware queues run out the
    remaining shards share (a warning says so)."""
    cur = torch.cuda.current_stream(device)
    chosen, spare = [], []
    for _ in range(candidates):
        if len(chosen) == n:
            break
        s = torch.cuda.Stream(device=device)
        (chosen if all(streams_overlap(s, o) for o in [cur] + chosen) else spare).append(s)
    if len(chosen) < n:
        warnings.warn(f"only {len(chosen)} of {n} stream shards get a hardware queue of their own; the rest share")
        chosen += spare[: n - len(chosen)]
        while len(chosen) < n:
            chosen.append(torch.cuda.Stream(device=device))
    return chosen


class StreamShards:
    """Intra-GPU sharding of the continual stream axis over HIP streams.

    At ~1000 streams a block's launch has ~800 workgroups for 512 resident slots (1.56 rounds), so the second
    round runs 56 % full.  Streams are independent, so the stream axis is cut into ``n_shards`` contiguous
    shards, each with its own model instance / state slab, advanced on its own HIP stream: the partially filled
    tail round of one shard's launch overlaps the other shards' launches.  Results are identical to one big
    shard (no cross-stream arithmetic).  ``make_model`` builds one (already-on-device, eval) CoStGcn."""

    def __init__(self, make_model, n_streams: int, n_shards: int, device):
        self.bounds = [shard_bounds(n_streams, r, n_shards) for r in range(n_shards)]
        self.models = [make_model() for _ in range(n_shards)]
        self.streams = concurrent_streams(n_shards, device) if n_shards > 1 else [None]
        self.device = device

    def forward_cycle(self, frames):
        """frames: list of (N, C, V, M) tensors -> concatenated logits of the last emission (or None)."""
        if len(self.models) == 1:
            outs = self.models[0].forward_cycle(frames)
            return outs[-1] if outs else None
        cur = torch.cuda.current_stream(self.device)
        parts = []
        for (lo, hi), model, st in zip(self.bounds, self.models, self.streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs = model.forward_cycle([f[lo:hi] for f in frames])
                parts.append(outs[-1] if outs else None)
        for st in self.streams:
            cur.wait_stream(st)
        return None if any(p is None for p in parts) else torch.cat(parts, dim=0)

    def state_bytes(self):
        return sum(m.state_bytes() for m in self.models)

    def scratch_bytes(self):
        return sum(m.scratch_bytes() for m in self.models)
