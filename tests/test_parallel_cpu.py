"""world_size-2 and world_size-8 gloo tests of the batch-shard + logit all-gather path (runs on CPU)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _bootstrap


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = _bootstrap.load()
    from continual_skeletons_amd import parallel
    full = torch.arange(n_total * 5, dtype=torch.float32).view(n_total, 5)   # "logits" every rank can recompute
    lo, hi = parallel.shard_bounds(n_total, rank, world)
    got = parallel.all_gather_ragged(full[lo:hi].clone(), n_total)
    ok = torch.equal(got, full)
    if n_total % world == 0:
        ok = ok and torch.equal(parallel.all_gather_logits(full[lo:hi].clone()), full)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def _run(n_total, world=2):
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = dict(q.get(timeout=300) for _ in procs)
    finally:
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.kill()                                  # the exact processes this test started
    assert res == {r: True for r in range(world)}


def test_shard_bounds_cover_and_partition():
    pkg = _bootstrap.load()
    from continual_skeletons_amd import parallel
    for n, w in [(8192, 8), (10, 3), (7, 8), (1, 2)]:
        b = [parallel.shard_bounds(n, r, w) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_all_gather_even_world2():
    _run(8)


def test_all_gather_ragged_world2():
    _run(7)


def test_all_gather_even_world8():
    """BASELINE configs[4] has 8 ranks: the rank-major order of the gathered logits at world size 8 (64 clips, 8 per rank;
    the 8-GPU node itself is the driver's to run)."""
    _run(64, world=8)


def test_all_gather_ragged_world8():
    """Uneven shards at 8 ranks: 61 clips -> five ranks of 8 and three of 7 (shard_bounds), and fewer clips than ranks (5)."""
    _run(61, world=8)
    _run(5, world=8)


def test_bench_self_launches_its_ranks_without_a_gpu_call_in_the_parent():
    """`python bench.py --gpus 2` (no launcher, no RANK in the environment): the parent must start the rank processes
    itself instead of exiting with a usage message.  On this CPU box the ranks then fail for lack of a GPU -- what is
    checked is that they were started by torch.distributed.run, that the parent relayed their failure as a non-zero exit
    code and that it printed nothing on stdout (no half-made JSON line)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-cpu-baseline", "--batch", "2"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=600)
    if out.returncode == 0:                      # a box with >= 2 GPUs: the line must be the 2-rank one
        import json
        assert json.loads(out.stdout.strip().splitlines()[-1])["n_gpus"] == 2
        return
    assert "launch multi-GPU runs with" not in out.stderr
    assert "local_rank: 0" in out.stderr or "ChildFailedError" in out.stderr, out.stderr[-1500:]
    assert out.stdout.strip() == ""
