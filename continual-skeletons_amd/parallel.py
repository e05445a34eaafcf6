"""Batch-axis sharding over the GPUs of one node (BASELINE.json configs[4]).

Every clip / stream is independent through all ten blocks (eval-mode BN is a per-channel affine, no
cross-sample op), so the N axis is partitioned into contiguous slices, weights are replicated
(12.6 MB) and continual state stays with its streams.  The one exchange step is an all-gather of the
logits ``(N/world, classes)`` -- RCCL over xGMI when the process group is "nccl"; 240 KiB per rank at
1024 clips/GPU, i.e. latency-bound, so it is issued as a single in-place collective on the compute
stream.  The two skeletons (M) of a clip never split: sharding is over whole clips.
"""
import torch
import torch.distributed as dist


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
