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
           "--steps", "2", "--warmup", "1", "--step-cycles", "2", "--stream-shards", "2", "--no-cpu-baseline",
           "--config5-batch", "4"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 6
    # the configs[4] leg (1024 clips / GPU in a real run) beside the weak-scaling headline; two gloo ranks SHARE the GPU here
    assert d["config5"]["clips_per_gpu"] == 4 and d["config5"]["global_batch"] == 8 and d["config5"]["value"] > 0
    assert d["ranks_seen"] == 1 and d["collective_backend"] == "gloo"
    assert d["value"] > 0 and d["costgcn_online"]["value"] > 0 and d["costgcn_online"]["streams_per_gpu"] == 6
    assert "agcn_kinetics" not in d          # the config-4 side numbers are per GPU, reported at N = 1 only
    # whole-job work is priced against n_gpus x one GPU's peak in every leg: no fraction can reach 1 (round-4 review: at
    # N = 8 the accounting would have printed ~6.0)
    for rc in (d["roofline_config"], d["config5"]["roofline_config"], d["costgcn_online"]["roofline_config"]):
        assert rc["n_gpus"] == 2 and 0 < rc["frac"] < 1 and 0 < rc["frac_alg"] < 1, rc
    assert 0 < d["costgcn_online"]["throughput_mode"]["roofline_config_frac"] < 1
    assert 0 < d["roofline"]["frac"] < 1 and 0 < d["config5"]["roofline"]["frac"] < 1
    # one row per rank with the device it ran on and its own clock; the two gloo ranks share the one GPU here
    pr = d["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1] and all(r["ms_per_step"] > 0 and r["device_uuid"] for r in pr)
    assert len({r["device_uuid"] for r in pr}) == d["ranks_seen"] == 1
    assert max(r["ms_per_step"] for r in pr) <= d["ms_per_step"] * 1.001 + 1e-3


def test_bench_self_launch_two_ranks_toy_sizes():
    """`python bench.py --gpus 2` WITHOUT a launcher: the parent (no GPU call of its own) starts the two ranks itself and relays
    rank 0's line -- the form the driver's scaling run uses (`python3 bench.py --gpus N ...`)."""
    env = dict(os.environ, CSK_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "4", "--streams", "6", "--steps", "2",
           "--warmup", "1", "--step-cycles", "2", "--stream-shards", "2", "--no-cpu-baseline", "--config5-batch", "3",
           "--no-split-leg"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]          # ONE line on stdout, nothing else
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 8
    assert d["config5"]["clips_per_gpu"] == 3 and d["config5"]["global_batch"] == 6 and d["config5"]["value"] > 0
    assert d["ranks_seen"] == 1 and d["collective_backend"] == "gloo" and d["value"] > 0
    assert len(d["per_rank"]) == 2 and 0 < d["roofline_config"]["frac"] < 1 and d["roofline_config"]["n_gpus"] == 2


def test_bench_self_launch_eight_ranks_toy_sizes():
    """The rank count of BASELINE configs[4]: `python bench.py --gpus 8` (self-launch, gloo, the eight ranks SHARE the one GPU
    of this box) must produce ONE line with eight `per_rank` rows, whole-job fractions priced against 8 x one GPU's peak
    (0 < frac < 1) and the configs[4] leg over 8 shards.  The RCCL exchange over xGMI itself can only run on the driver's
    8-GPU node (README: no scaling curve has been measured)."""
    env = dict(os.environ, CSK_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--batch", "2", "--streams", "4", "--steps", "2",
           "--warmup", "1", "--step-cycles", "2", "--stream-shards", "1", "--no-cpu-baseline", "--config5-batch", "2",
           "--no-split-leg"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["global_batch"] == 16
    assert d["config5"]["clips_per_gpu"] == 2 and d["config5"]["global_batch"] == 16 and d["config5"]["value"] > 0
    assert d["ranks_seen"] == 1 and d["collective_backend"] == "gloo" and d["value"] > 0
    pr = d["per_rank"]
    assert [r["rank"] for r in pr] == list(range(8)) and all(r["ms_per_step"] > 0 and r["device_uuid"] for r in pr)
    for rc in (d["roofline_config"], d["config5"]["roofline_config"], d["costgcn_online"]["roofline_config"]):
        assert rc["n_gpus"] == 8 and 0 < rc["frac"] < 1 and 0 < rc["frac_alg"] < 1, rc


RCCL_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["CSK_ROOT"])
import torch, torch.distributed as dist
import _bootstrap, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)          # "nccl" IS RCCL on ROCm
pkg = _bootstrap.load()
from continual_skeletons_amd import parallel
net = pkg.StGcn(pkg.ntu_graph().A, input_shape=(3, 40, 25, 2), num_classes=60).eval()
bench.randomise_(net, seed=0)
net = net.to(dev)
x = torch.rand((5, 3, 40, 25, 2), generator=torch.Generator().manual_seed(11)).to(dev)
ok = True
for it in range(3):                       # stream ordering: the collective follows the kernels that produce its input
    logits = net(x)
    g1 = parallel.all_gather_logits(logits)
    g2 = parallel.all_gather_ragged(logits, 5)
    torch.cuda.synchronize()
    ok = ok and torch.equal(g1, logits) and torch.equal(g2, logits) and bool(torch.isfinite(g1).all())
# the side stream of a StreamShards engine next to RCCL's own streams (hardware-queue probe, parallel.concurrent_streams)
def make():
    co = pkg.CoStGcn(pkg.ntu_graph().A, pool_size=3, pool_padding=1).eval()
    bench.randomise_(co, seed=0)
    return co.to(dev)
eng = parallel.StreamShards(make, 6, 2, dev)
frames = torch.rand((92, 6, 3, 25, 2), generator=torch.Generator().manual_seed(12)).to(dev)
seen = 0
for c in range(23):
    out = eng.forward_cycle([frames[4 * c + f] for f in range(4)])
    if out is not None:
        g = parallel.all_gather_logits(out)
        torch.cuda.synchronize()
        ok = ok and torch.equal(g, out)
        seen += 1
t = torch.tensor([1.0], device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
print(f"RCCL backend={dist.get_backend()} ok={ok} seen={seen} allreduce={float(t.item())}", flush=True)
dist.destroy_process_group()
sys.exit(0 if (ok and seen > 0) else 1)
'''


def test_rccl_world_size_1_all_gather_on_the_gpu(tmp_path):
    """As close to RCCL as one GPU allows: a real ``backend="nccl"`` (= RCCL) process group of world size 1 bound to cuda:0
    (library load, communicator init with ``device_id``, stream ordering between our HIP launches and the collective,
    the probed shard streams beside RCCL's own): real logits through parallel.all_gather_logits / all_gather_ragged and
    the all_reduce + barrier bench.py uses.  The 8-rank exchange itself can only run on the driver's 8-GPU node."""
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), CSK_ROOT=ROOT, RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, f"{out.stdout[-1000:]} {out.stderr[-3000:]}"
    assert "RCCL backend=nccl ok=True" in out.stdout


def test_bench_one_rank_rccl_toy_sizes():
    """bench.py --gpus 1 under torch.distributed.run with the default backend: WORLD_SIZE = 1 takes the single-process
    path; with CSK_BENCH_FORCE_DIST=1 it initialises RCCL anyway and runs every collective of the N > 1 path (barrier,
    max-over-ranks all_reduce, logit all-gather, config5 leg) on one rank."""
    env = dict(os.environ, CSK_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "3", "--streams", "6",
           "--steps", "2", "--warmup", "1", "--step-cycles", "2", "--stream-shards", "2", "--no-cpu-baseline",
           "--config5-batch", "4"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["ranks_seen"] == 1 and d["collective_backend"] == "nccl"
    assert d["config5"]["clips_per_gpu"] == 4 and d["config5"]["value"] > 0
    assert len(d["per_rank"]) == 1 and d["per_rank"][0]["rank"] == 0 and "error" not in d


def test_bench_fails_the_line_when_rccl_ranks_share_a_device():
    """Two RCCL ranks pinned to ONE device (a launcher that ignores LOCAL_RANK): RCCL refuses the communicator -- or, if it
    ever came up, bench.py marks the line invalid (`error`, ranks_seen != n_gpus) and exits non-zero.  Either way no valid
    2-GPU line may come out of a 1-GPU box."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CSK_BENCH_SAME_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "2", "--streams", "4",
           "--steps", "1", "--warmup", "1", "--step-cycles", "1", "--no-cpu-baseline", "--workload", "clip", "--no-split-leg",
           "--config5-batch", "2"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    for ln in out.stdout.splitlines():
        if ln.startswith("{"):
            d = json.loads(ln)
            assert "error" in d and d["ranks_seen"] != d["n_gpus"]
