"""N > 1 ranks of the sharded path on hardware, as far as a 1-GPU box allows: two gloo ranks SHARE cuda:0 (RCCL refuses
two ranks per device, so the collective itself runs over gloo here; the data path -- shard_bounds slices through the
HIP kernels, ragged all-gather, comparison with the unsharded result -- is the one the 8-GPU run uses).

(i)  each rank runs its slice of a 6-clip batch through StGcn and its slice of an 8-stream slab through
     CoStGcn.forward_cycle; the gathered logits must equal the single-process result BITWISE on every rank;
(ii) bench.py itself under `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2` at toy sizes
     (CSK_BENCH_BACKEND=gloo): one JSON line with n_gpus = 2 and the global batch of both ranks.
The launcher / parent process never touches the GPU before it starts its children (no fork / exec after HIP init).
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["CSK_ROOT"])
import torch, torch.distributed as dist
import _bootstrap, bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
pkg = _bootstrap.load()
from continual_skeletons_amd import parallel
dev = torch.device("cuda:0")
A = pkg.ntu_graph().A
# ---- clip: 6 clips (odd split for world 4, even for 2), T = 40
net = pkg.StGcn(A, input_shape=(3, 40, 25, 2), num_classes=60).eval()
bench.randomise_(net, seed=0)                       # identical weights on every rank
net = net.to(dev)
x = torch.rand((6, 3, 40, 25, 2), generator=torch.Generator().manual_seed(11)).to(dev)
lo, hi = parallel.shard_bounds(6, rank, world)
mine = net(x[lo:hi].contiguous())
gathered = parallel.all_gather_ragged(mine, 6)
full = net(x)
ok_clip = bool(torch.equal(gathered, full))
if not ok_clip:
    print(f"RANK{rank} clip: local-vs-full-slice max diff {float((mine - full[lo:hi]).abs().max())}, "
          f"gathered-vs-full max diff {float((gathered - full).abs().max())}", flush=True)
# ---- continual: 8 streams (ragged 7 as well), cycles of 4 frames until predictions appear
ok_step = True
for n_streams in (8, 7):
    frames = torch.rand((96, n_streams, 3, 25, 2), generator=torch.Generator().manual_seed(12)).to(dev)
    def make():
        co = pkg.CoStGcn(A, pool_size=4, pool_padding=1).eval()
        bench.randomise_(co, seed=0)
        return co.to(dev)
    lo, hi = parallel.shard_bounds(n_streams, rank, world)
    part, whole = make(), make()
    seen = 0
    for c in range(24):
        fr = [frames[4 * c + f] for f in range(4)]
        a = part.forward_cycle([f[lo:hi].contiguous() for f in fr])
        b = whole.forward_cycle(fr)
        assert len(a) == len(b)
        for la, lb in zip(a, b):
            g = parallel.all_gather_ragged(la, n_streams)
            if not torch.equal(g, lb):
                print(f"RANK{rank} step n={n_streams} cycle {c}: local-vs-slice {float((la - lb[lo:hi]).abs().max())}, "
                      f"gathered-vs-whole {float((g - lb).abs().max())}", flush=True)
                ok_step = False
            seen += 1
    ok_step = ok_step and seen > 0
print(f"RANK{rank} clip={ok_clip} step={ok_step}", flush=True)
dist.destroy_process_group()
sys.exit(0 if (ok_clip and ok_step) else 1)
'''


def test_two_ranks_share_the_gpu_sharded_equals_unsharded(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CSK_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for rank, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank}: {so[-500:]} {se[-2000:]}"
        assert f"RANK{rank} clip=True step=True" in so


def test_bench_two_ranks_toy_sizes():
    env = dict(os.environ, CSK_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "3", "--streams", "6",
           "--steps", "2", "--warmup", "1", "--step-cycles", "2", "--stream-shards", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 6
    assert d["value"] > 0 and d["costgcn_online"]["value"] > 0 and d["costgcn_online"]["streams_per_gpu"] == 6
    assert "agcn_kinetics" not in d          # the config-4 side numbers are per GPU, reported at N = 1 only
