#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X-native ST-GCN / CoST-GCN forward path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload clip|step] [--batch B]

One "step" = one pass of the hot path over one batch of synthetic input already resident in HBM:
  clip : full 10-block ST-GCN clip forward (input norm -> 10 blocks -> pool+fc), batch 256 clips per GPU,
         NTU-60 shape (3, 300, 25, 2), fp32                                  [BASELINE.json configs[1]]
  step : CoST-GCN online inference, one new frame for each of 1024 concurrent streams per GPU, persistent
         ring-buffer state in HBM                                              [BASELINE.json configs[2]]
Multi-GPU (torchrun, one process per GPU): the batch / stream axis is sharded (weak scaling: per-GPU work is
fixed), the only exchange is one RCCL all-gather of logits per step.  Rank 0 prints ONE JSON line.

`roofline` is computed for the dominant kernel (tcn_stage_kernel: 9x1 temporal conv + BN + residual + ReLU,
70 % of the path's FLOPs): algorithmic FLOPs of its launches (SURVEY 8d accounting) / their duration measured
live with HIP events on the launch stream, against the dense fp32 MFMA peak (157.3 TFLOP/s).
`cpu_baseline` times the CPU oracle (a port of the reference's op sequence, see oracle/) on the host cores.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tools"))
import workmodel  # noqa: E402  (tools/workmodel.py: SURVEY 8d work accounting, shared with the profile summarisers)

PEAK_F32_MFMA_TFLOPS = workmodel.PEAK_F32_MFMA_TFLOPS     # MI355X_MICROARCH.md: dense f32 MFMA (v_mfma_f32_32x32x2_f32)
NTU = dict(C=3, T=300, V=25, M=2, classes=60)
KIN_SHAPE = (3, 300, 18, 2)        # Kinetics-400 skeleton shape (C, T, V, M), BASELINE.json configs[3]


def randomise_(net, seed, attn_scale=1.0):
    """Random-init weights of the named architecture with non-trivial BN statistics (data: synthetic).  EVERY parameter
    and buffer is drawn from the seeded generator (the constructors' own kaiming / normal inits use the unseeded global
    RNG and would differ from rank to rank and from instance to instance).  ``attn_scale``: graph_attn scale -- 1 for
    ST-GCN (multiplicative mask, base.py:262), ~1/V for A-GCN where it is an additive dense matrix (a_gcn.py:50)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, prm in net.named_parameters():
            if name.endswith(".A") or name == "A":
                continue                                                   # the graph's adjacency
            if name.endswith("graph_attn"):
                prm.copy_((torch.rand(prm.shape, generator=g) + 0.5) * attn_scale)
            elif "bn" in name and name.endswith("weight"):
                prm.copy_(torch.rand(prm.shape, generator=g) * 0.5 + 0.25)
            elif name.endswith("bias"):
                prm.copy_(torch.rand(prm.shape, generator=g) * 0.2 - 0.1)
            elif name.endswith("weight") and prm.dim() >= 2:              # conv / linear: kaiming-like, fan-in scaled
                fan_in = prm[0].numel()
                prm.copy_(torch.randn(prm.shape, generator=g) * (1.0 / fan_in) ** 0.5)
        for name, buf in net.named_buffers():
            if name.endswith("running_var"):
                buf.copy_(torch.rand(buf.shape, generator=g) + 0.5)
            elif name.endswith("running_mean"):
                buf.copy_(torch.rand(buf.shape, generator=g) * 0.2 - 0.1)


class LaunchTimer:
    """HIP-event timing of every launch of one stage, on the stream the kernels are launched on."""

    def __init__(self, pkg, name):
        self.pkg, self.name, self.orig = pkg, name, getattr(pkg.blocks, name)
        self.records, self.enabled = [], False

    def __enter__(self):
        def wrapped(*a, **k):
            if not self.enabled:
                return self.orig(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = self.orig(*a, **k)
            e1.record()
            self.records.append((e0, e1))
            return out
        setattr(self.pkg.blocks, self.name, wrapped)
        return self

    def __exit__(self, *exc):
        setattr(self.pkg.blocks, self.name, self.orig)

    def total_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.records)


def tcn_flops_per_clip_forward(nm, c_in=3, T=300, V=25):
    """Algorithmic FLOPs of the ten tcn_stage launches for nm skeleton sequences (TCN 9x1 +
    block-residual 1x1 MACs, 2 FLOP per MAC)."""
    from continual_skeletons_amd.models import layer_table
    t, macs = T, 0
    for (ci, co, s, res) in layer_table(c_in):
        t_out = (t + 8 - 9) // s + 1
        per_pos = 9 * co * co + (ci * co if (res and (ci != co or s != 1)) else 0)
        macs += per_pos * t_out * V
        t = t_out
    return 2 * macs * nm


def host_cpu_threads(cap=16):
    """Threads for both CPU baseline legs: the CPUs this process may really use -- the smaller of its affinity mask and
    its cgroup CPU quota (a GPU box's container exposes 256 logical CPUs under a 16-CPU quota: an OpenMP team sized by
    the mask would spin for minutes) -- and at most `cap`, where the oracle's torch-CPU ops stop scaling anyway."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def _median_rate(fn, units, runs=5, warm=2, budget_s=10.0):
    """Median of `runs` timed calls after `warm` untimed ones (SURVEY 8d / BASELINE.md 4 protocol), inside a time
    budget: the first (warm-up) call is timed and the number of further calls is cut to what the budget allows (at
    least one timed call).  Returns (units per second, runs timed)."""
    import statistics
    t0 = time.perf_counter()
    fn()
    first = time.perf_counter() - t0
    afford = int(max(0.0, budget_s - first) / max(first, 1e-6))
    if afford >= runs + warm - 1:
        for _ in range(warm - 1):
            fn()
    runs = max(1, min(runs, afford))
    ts = []
    for _ in range(runs):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return units / statistics.median(ts), len(ts)


def cpu_baseline_clip(seed, threads, budget_s=30.0, adaptive=False):
    """Oracle (CPU port of the reference op sequence) per the reference's own protocol (scripts/benchmark_all_ntu60.py:
    15-18,52: batch 1; plus batch 8): median of 5 after 2 warm-ups, at `threads` threads and (batch 1) at one thread,
    each leg inside its share of the time budget.  adaptive: A-GCN at the Kinetics-400 shape (configs[3])."""
    from oracle import stgcn_oracle as o
    import _bootstrap
    pkg = _bootstrap.load()
    if adaptive:
        net = pkg.AGcn(pkg.kinetics_graph().A, KIN_SHAPE, 400).eval()
        randomise_(net, seed, attn_scale=1 / 18)
        shape, fwd, what = KIN_SHAPE, (lambda xx, sd: o.stgcn_forward(xx, sd, gcn=o.adaptive_graph_conv)), "A-GCN, Kinetics-400-shape clips"
    else:
        net = pkg.StGcn(pkg.ntu_graph().A).eval()
        randomise_(net, seed)
        shape, fwd, what = (NTU["C"], NTU["T"], NTU["V"], NTU["M"]), o.stgcn_forward, "NTU-60 clips"
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    x = torch.rand((8,) + tuple(shape), generator=torch.Generator().manual_seed(1))
    runs = {}
    with torch.no_grad():
        torch.set_num_threads(threads)
        runs["batch8"], n8 = _median_rate(lambda: fwd(x, sd), 8, budget_s=0.4 * budget_s)
        runs["batch1"], n1 = _median_rate(lambda: fwd(x[:1], sd), 1, budget_s=0.2 * budget_s)
        torch.set_num_threads(1)
        runs["batch1_one_thread"], n1t = _median_rate(lambda: fwd(x[:1], sd), 1, runs=3, warm=1, budget_s=0.4 * budget_s)
        torch.set_num_threads(threads)
    best = max(runs["batch8"], runs["batch1"])
    return dict(value=round(best, 3), unit="clips/s", cores=threads, kind="port",
                runs={k: round(v, 3) for k, v in runs.items()},
                sample=f"oracle.stgcn_forward, torch CPU fp32, {what}; median of {n8} / {n1} / {n1t} timed passes "
                       f"(batch 8 and batch 1 at {threads} threads, batch 1 at 1 thread) after warm-up; value = best of the {threads}-thread runs")


def run_step_workload(pkg, dev, streams, cycles, warm_cycles, rank, world, parallel, dist, shards=1, native_plan=True, fpl=4, use_dist=None):
    """CoST-GCN online inference: `streams` concurrent streams per GPU, persistent ring-buffer state; one
    cycle = 4 consecutive frames (the stack's stride pattern) = one prediction per stream.  With shards > 1 the
    stream axis is split into independent shards advanced on separate HIP streams (parallel.StreamShards)."""
    use_dist = world > 1 if use_dist is None else use_dist

    def make():
        net = pkg.CoStGcn(pkg.ntu_graph().A).eval()
        net.use_native_plan = native_plan     # False: launches are driven from Python so that they can be timed one by one
        net.set_max_cycle(min(8, max(4, fpl)))   # rings sized for the cycles this leg issues (warm-up runs single frames)
        randomise_(net, seed=0)
        return net.to(dev)
    eng = parallel.StreamShards(make, streams, shards, dev)
    g = torch.Generator(device=dev).manual_seed(200 + rank)
    frames = torch.rand((8, streams, NTU["C"], NTU["V"], NTU["M"]), device=dev, generator=g)   # resident inputs
    for t in range(76 + 4 * (75 - 19 - 1)):     # 76 warm-up frames (models/base.py:144-159) + fill the temporal pool
        eng.forward_cycle([frames[t % 8]])
    fi = 0

    def cycle():
        nonlocal fi
        out = eng.forward_cycle([frames[(fi + f) % 8] for f in range(fpl)])   # fpl frames, one launch pair per block
        fi += fpl
        return parallel.all_gather_logits(out) if (use_dist and out is not None) else out

    with LaunchTimer(pkg, "tcn_step_launch") as lt:
        for _ in range(warm_cycles):
            out = cycle()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        lt.enabled = shards == 1            # per-launch events are only meaningful without concurrent shards
        t0 = time.perf_counter()
        for _ in range(cycles):
            out = cycle()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        lt.enabled = False
        tcn_ms, n_launch = lt.total_ms(), len(lt.records)
    assert out is not None and out.shape == (streams * world, NTU["classes"]) and bool(torch.isfinite(out).all())
    return dt, tcn_ms, n_launch, (eng.state_bytes(), eng.scratch_bytes())


def cpu_baseline_step(seed, threads, budget_s=20.0, adaptive=False):
    """Oracle continual path (port of the reference op sequence + restated continual protocol), one stream (batch 1, the
    reference's protocol): 76 warm-up frames (models/base.py:144-159), then the median rate of up to 5 segments of
    steady-state frames (100 each when the time budget allows) -- at `threads` threads and at one thread.
    adaptive: CoAGCN (per-frame attention) at the Kinetics-400 shape (configs[3])."""
    from oracle import stgcn_oracle as o
    import _bootstrap
    import statistics
    pkg = _bootstrap.load()
    if adaptive:
        net = pkg.CoAGcn(pkg.kinetics_graph().A, KIN_SHAPE, 400).eval()
        randomise_(net, seed, attn_scale=1 / 18)
        shape, what = KIN_SHAPE, "CoAGCN, one Kinetics-400-shape stream"
    else:
        net = pkg.CoStGcn(pkg.ntu_graph().A).eval()
        randomise_(net, seed)
        shape, what = (NTU["C"], NTU["T"], NTU["V"], NTU["M"]), "one NTU-60 stream"
    sd = {k.replace("0.1.", "").replace("0.0.residual", "residual"): v.clone() for k, v in net.state_dict().items()}
    x = torch.rand((1, shape[0], 200, shape[2], shape[3]), generator=torch.Generator().manual_seed(2))
    info = {}

    def rate(tag, nthreads, budget):
        torch.set_num_threads(nthreads)
        orc = o.CoStGcnOracle(sd)
        if adaptive:
            for b in orc.blocks:
                b.gcn = o.adaptive_graph_conv
        rates = []
        with torch.no_grad():
            t0 = time.perf_counter()
            for t in range(76):
                orc.forward_step(x[:, :, t])
            warm = time.perf_counter() - t0                      # early frames are cheaper (upper blocks idle): a lower bound
            per_frame = max(warm / 76 * 1.5, 1e-5)
            seg_frames = int(min(100, max(8, (budget - warm) / 5 / per_frame)))
            segments = 5 if (budget - warm) / per_frame >= 5 * seg_frames else 3
            n = 0
            for _ in range(segments):
                t0 = time.perf_counter()
                for _ in range(seg_frames):
                    orc.forward_step(x[:, :, 76 + n % 124])
                    n += 1
                rates.append(seg_frames / (time.perf_counter() - t0))
        info[tag] = (segments, seg_frames)
        return statistics.median(rates)

    runs = {"all_threads": rate("all_threads", threads, 0.6 * budget_s), "one_thread": rate("one_thread", 1, 0.4 * budget_s)}
    torch.set_num_threads(threads)
    best = max(runs, key=runs.get)          # tiny per-frame ops: one thread can beat the thread pool
    return dict(value=round(runs[best], 2), unit="frames/s", cores=threads if best == "all_threads" else 1, kind="port",
                runs={k: round(v, 2) for k, v in runs.items()},
                sample=f"oracle.CoStGcnOracle, {what}: 76 warm-up frames, then median of {info['all_threads'][0]} segments "
                       f"of {info['all_threads'][1]} steady-state frames at {threads} threads ({info['one_thread'][0]} x {info['one_thread'][1]} "
                       f"at 1 thread); value = the faster of the two")


def run_config4(pkg, dev, parallel, batch=64, streams=1024, shards=2, cpu_threads=0, cpu_budget=14.0):
    """BASELINE.json configs[3]: A-GCN (per-sample adaptive adjacency) clip forward and CoAGCN online step at the
    Kinetics-400 shape (V = 18, T = 300), synthetic inputs resident in HBM, random-init weights.  Both legs carry the
    whole-config roofline (tools/workmodel.py with the A-GCN terms: embedding convs, attention logits, dense
    aggregation -- all of it executed, so frac == frac_executed) and, when cpu_threads > 0, the CPU oracle beside them."""
    A, shape = pkg.kinetics_graph().A, KIN_SHAPE
    V, M = shape[2], shape[3]
    cpu_clip = cpu_step = None
    if cpu_threads > 0:
        cpu_clip = cpu_baseline_clip(0, cpu_threads, budget_s=0.55 * cpu_budget, adaptive=True)
        cpu_step = cpu_baseline_step(0, cpu_threads, budget_s=0.45 * cpu_budget, adaptive=True)
    net = pkg.AGcn(A, shape, 400).eval()
    randomise_(net, 0, attn_scale=1 / 18)
    net = net.to(dev)
    x = torch.rand((batch,) + shape, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
    for _ in range(2):
        out = net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = net(x)
    torch.cuda.synchronize()
    clip_dt = (time.perf_counter() - t0) / 5
    assert out.shape == (batch, 400) and bool(torch.isfinite(out).all())
    # the same forward in the opt-in bf16x3 precision mode (temporal convs on the bf16 matrix pipe; the adaptive graph conv and
    # the attention stay exact fp32): reported under its own key, never as the leg's value
    pkg.set_precision(net, "bf16x3")
    for _ in range(2):
        out3 = net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out3 = net(x)
    torch.cuda.synchronize()
    clip3_dt = (time.perf_counter() - t0) / 5
    clip3_diff = float((out3 - out).abs().max())
    assert bool(torch.isfinite(out3).all())
    del net, x, out, out3

    def make():
        co = pkg.CoAGcn(A, shape, 400).eval()
        co.set_max_cycle(4)                      # this leg issues 4-frame cycles
        randomise_(co, 0, attn_scale=1 / 18)
        return co.to(dev)
    eng = parallel.StreamShards(make, streams, shards, dev)
    frames = torch.rand((8, streams) + (3, V, M), device=dev, generator=torch.Generator(device=dev).manual_seed(8))
    for t in range(76 + 4 * (75 - 19 - 1)):       # warm-up + fill the temporal pool, as the CoST-GCN leg
        eng.forward_cycle([frames[t % 8]])
    for c in range(2):
        eng.forward_cycle([frames[(4 * c + f) % 8] for f in range(4)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for c in range(12):
        out = eng.forward_cycle([frames[(4 * c + f) % 8] for f in range(4)])
    torch.cuda.synchronize()
    step_dt = (time.perf_counter() - t0) / 12
    assert out is not None and out.shape == (streams, 400) and bool(torch.isfinite(out).all())
    sbytes = eng.state_bytes()
    scratch = eng.scratch_bytes()
    del eng
    gc.collect()
    torch.cuda.empty_cache()
    cfa, cfe, cby = workmodel.clip_totals(batch * M, V=V, adaptive=True)
    sfa, sfe, sby = workmodel.step_totals(streams * M, 4, V=V, adaptive=True)
    return {"config": "BASELINE.json configs[3], Kinetics-400 shape (3,300,18,2), fp32, synthetic, per GPU",
            "agcn_clip": {"value": round(batch / clip_dt, 1), "unit": "clips/s", "batch": batch, "ms_per_step": round(clip_dt * 1e3, 3),
                          "roofline_config": workmodel.roofline_config(cfa, cby, clip_dt, cfe), "cpu_baseline": cpu_clip,
                          "bf16x3": {"value": round(batch / clip3_dt, 1), "unit": "clips/s", "ms_per_step": round(clip3_dt * 1e3, 3),
                                     "dtype": "bf16x3-split temporal convs, f32 accumulate; graph conv / attention exact f32",
                                     "speedup_vs_f32": round(clip_dt / clip3_dt, 3), "max_abs_logit_diff_vs_f32": clip3_diff}},
            "coagcn_online": {"value": round(4 * streams / step_dt, 1), "unit": "frames/s", "streams": streams,
                              "stream_shards": shards, "frames_per_launch": 4, "ms_per_frame_step": round(step_dt / 4 * 1e3, 4),
                              "state_slab_GB_per_gpu": round(sbytes / 1e9, 3), "split_k_scratch_GB_per_gpu": round(scratch / 1e9, 3),
                              "roofline_config": workmodel.roofline_config(sfa, sby, step_dt, sfe), "cpu_baseline": cpu_step}}


def run_latency(pkg, dev, cpu_step, stream_counts=(1, 16), frames=400):
    """Few-stream latency of the online path (the reference's own CPU protocol is batch 1: scripts/benchmark_all_ntu60.py:17):
    one frame at a time through CoStGcn.forward_step in latency mode (set_latency_mode: split-K on launches too small to
    fill the GPU), host-synchronised after every frame so that a sample is the whole submit -> result time of a frame.
    p50 / p99 over `frames` steady-state frames; `pipelined_ms_per_frame` = the same frames without per-frame
    synchronisation (throughput of a single stream set)."""
    import statistics
    out = []
    for streams in stream_counts:
        net = pkg.CoStGcn(pkg.ntu_graph().A).eval()
        randomise_(net, seed=0)
        net = net.to(dev)
        net.set_latency_mode(8)
        x = torch.rand((8, streams, NTU["C"], NTU["V"], NTU["M"]), device=dev, generator=torch.Generator(device=dev).manual_seed(5))
        for t in range(76 + 4 * 56):               # warm-up + fill the temporal pool: every frame class (4 phases) is steady
            net.forward_step(x[t % 8])
        torch.cuda.synchronize()
        ts, preds = [], 0
        for t in range(frames):
            t0 = time.perf_counter()
            o = net.forward_step(x[t % 8])
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            preds += o is not None
        t0 = time.perf_counter()
        for t in range(frames):
            net.forward_step(x[t % 8])
        torch.cuda.synchronize()
        piped = (time.perf_counter() - t0) / frames
        ts.sort()
        # launches of one 4-frame stride cycle in per-frame stepping: input norm per frame; per block one graph-conv launch per
        # received frame and one temporal-conv launch per emission (each + its split-K reduction in latency mode); the head
        launches, recv = 4.0, 4.0
        for i in range(10):
            blk = net.layers[f"layer{i + 1}"]
            emit = recv / blk.stride
            launches += recv * (2 if blk._state.gcn_ksplit > 1 else 1) + emit * (2 if blk._state.ksplit > 1 else 1)
            recv = emit
        launches += 1                                      # the head: one launch on the predicting frame
        out.append({"streams": streams, "frames_timed": frames, "predictions": preds,
                    "ms_per_frame_p50": round(statistics.median(ts) * 1e3, 4), "ms_per_frame_p99": round(ts[int(0.99 * (frames - 1))] * 1e3, 4),
                    "ms_per_frame_mean": round(sum(ts) / frames * 1e3, 4), "pipelined_ms_per_frame": round(piped * 1e3, 4),
                    "launches_per_frame": round(launches / 4, 2),
                    "split_k": [net.layers[f"layer{i + 1}"]._state.ksplit for i in range(10)]})
        del net, x
        gc.collect()
        torch.cuda.empty_cache()
    cpu_ms = round(1e3 / cpu_step["value"], 4) if cpu_step and cpu_step.get("value") else None
    return {"mode": "CoStGcn.forward_step, one frame per call, set_latency_mode(8), synchronised after every frame",
            "per_stream_count": out, "cpu_oracle_ms_per_frame_one_stream": cpu_ms,
            "note": "frames 1-3 of a stride cycle run fewer blocks than the 4th (strides 2, 2): p99 is the predicting frame"}


def run_clip_latency(pkg, dev, cpu, batches=(1, 8), forwards=200, split_k=4):
    """Small-batch clip inference (the reference's own CPU protocol is batch 1: scripts/benchmark_all_ntu60.py:17): the whole
    StGcn forward at batch 1 and 8, every forward host-synchronised (a sample = submit -> logits ready), p50 / p99 over
    `forwards` forwards -- eager (Python launches every kernel) and as a replayed hipGraph (one launch per forward) -- in the
    default mode and in the latency mode (StGcn.set_latency_mode: split-K on every stage launch; the factor depends on the
    layer only, never on the batch).  `pipelined_ms` = graph replays back to back without per-forward synchronisation."""
    import statistics
    net = pkg.StGcn(pkg.ntu_graph().A, input_shape=(NTU["C"], NTU["T"], NTU["V"], NTU["M"]), num_classes=NTU["classes"]).eval()
    randomise_(net, seed=0)
    net = net.to(dev)

    def pct(ts, q):
        ts = sorted(ts)
        return round(ts[int(q * (len(ts) - 1))] * 1e3, 4)

    rows = []
    for b in batches:
        x = torch.rand((b, NTU["C"], NTU["T"], NTU["V"], NTU["M"]), device=dev, generator=torch.Generator(device=dev).manual_seed(40 + b))
        row = {"batch": b}
        ref = None
        for mode, sk in (("default", 0), ("latency_mode", split_k)):
            net.set_latency_mode(sk)
            for _ in range(3):
                out = net(x)
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            eager = []
            for _ in range(forwards):
                t0 = time.perf_counter()
                out = net(x)
                torch.cuda.synchronize()
                eager.append(time.perf_counter() - t0)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                gout = net(x)
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            graph = []
            for _ in range(forwards):
                t0 = time.perf_counter()
                g.replay()
                torch.cuda.synchronize()
                graph.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            for _ in range(forwards):
                g.replay()
            torch.cuda.synchronize()
            piped = (time.perf_counter() - t0) / forwards
            assert bool(torch.isfinite(gout).all()) and torch.equal(gout, out)
            row[mode] = {"split_k": sk, "eager_ms_p50": pct(eager, 0.5), "eager_ms_p99": pct(eager, 0.99),
                         "graph_ms_p50": pct(graph, 0.5), "graph_ms_p99": pct(graph, 0.99), "pipelined_ms": round(piped * 1e3, 4),
                         "clips_per_s_graph_p50": round(b / statistics.median(graph), 1),
                         "max_abs_logit_diff_vs_default": float((out - ref).abs().max())}
            del g, gout
        fa, fe, by = workmodel.clip_totals(b * NTU["M"])
        best = min(row["default"]["graph_ms_p50"], row["latency_mode"]["graph_ms_p50"])
        row["roofline_config_frac_best_graph_p50"] = workmodel.roofline_config(fa, by, best * 1e-3, fe)["frac"]
        rows.append(row)
        del x
    net.set_latency_mode(0)
    del net
    gc.collect()
    torch.cuda.empty_cache()
    cpu_b1 = cpu["runs"].get("batch1") if cpu else None
    return {"mode": "StGcn clip forward, NTU-60 shape, fp32, host-synchronised per forward", "forwards_timed": forwards,
            "per_batch": rows, "cpu_oracle_batch1_ms": round(1e3 / cpu_b1, 3) if cpu_b1 else None,
            "cpu_oracle_batch1_clips_per_s": cpu_b1}


def load_traffic(name="traffic_tcn_stage.json", batch=None, streams=None):
    """Per-launch HBM bytes of the dominant kernel from the COMMITTED PMC summaries (profiles/; separate rocprofv3 --pmc
    passes of tools/profile_r05.sh condensed by tools/summarize_pmc.py -- not measured by this run, the line says so under
    `traffic_source`).  The clip file holds one entry per profiled batch size (`by_batch`); returns the entry that matches
    `batch` / `streams`, or None when no pass was committed for that size."""
    p = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(p):
        return None
    with open(p) as f:
        d = json.load(f)
    if batch is not None:
        for e in d.get("by_batch", [d]):
            if e.get("batch") == batch:
                return e
        return None
    if streams is not None and d.get("streams") != streams:
        return None
    return d


def self_launch(n):
    """`python bench.py --gpus N` (N > 1) started WITHOUT a launcher: this process becomes the parent of N fresh rank
    processes (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`, one rank per
    GPU, rendezvous on 127.0.0.1), relays rank 0's single JSON line on stdout (anything else the ranks print goes to
    stderr) and returns the launcher's exit code.  The parent makes NO GPU call -- no torch.cuda.*, no _bootstrap.load(), no
    process group -- before or after starting its children, and it never re-executes itself: it only spawns and exits."""
    import signal
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL between processes needs it on this driver
    # --standalone: the launcher binds its own rendezvous port (no bind-close-rebind race with other processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(n), os.path.abspath(__file__)] + sys.argv[1:]
    # own session: the launcher and its N ranks form one process group that can be signalled as a whole, so that a driver
    # timeout that terminates this parent does not leave rank processes holding the GPUs
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)

    def forward(signum, _frame):
        try:
            os.killpg(proc.pid, signum)
        except ProcessLookupError:
            pass
    old_handlers = {sg: signal.signal(sg, forward) for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP)}
    lines = 0
    try:
        for ln in proc.stdout:
            if ln.startswith("{"):
                lines += 1
                if lines == 1:
                    sys.stdout.write(ln)
                    sys.stdout.flush()
                    continue
            sys.stderr.write(ln)
        rc = proc.wait()
    finally:
        if proc.poll() is None:                                # we are being torn down: take the ranks with us
            try:
                os.killpg(proc.pid, signal.SIGTERM)
                proc.wait(timeout=10)
            except (ProcessLookupError, subprocess.TimeoutExpired):
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
    if rc == 0 and lines != 1:
        sys.stderr.write(f"bench.py: the {n} ranks exited 0 but printed {lines} JSON lines (exactly one expected)\n")
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="both", choices=["both", "clip", "step"],
                    help="'both' (default): primary metric = clip, CoST-GCN online step reported in the same line")
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU")
    ap.add_argument("--streams", type=int, default=1024, help="concurrent CoST-GCN streams per GPU")
    ap.add_argument("--step-cycles", type=int, default=100,
                    help="timed launch cycles of the online workload (default 100 cycles of 4 frames = 400 timed frames per stream, SURVEY 8d)")
    ap.add_argument("--kernel-cycles", type=int, default=17,
                    help="cycles of the single-shard pass that times every tcn_step_kernel launch with HIP events (6 launches per cycle: 17 -> 102)")
    ap.add_argument("--stream-shards", type=int, default=1,
                    help="independent stream shards on separate HIP streams (rounds 2-5 ran 2: their tiles cut a launch into 800 k pieces "
                         "for 512 slots and a second shard filled the tail rounds; the slot-balanced tiles of csrc/step16.hip make every "
                         "launch of the 1024-stream cycle one full round, and one shard measures faster)")
    ap.add_argument("--frames-per-launch", type=int, default=4, help="frames advanced per launch cycle of the online workload")
    ap.add_argument("--config5-batch", type=int, default=1024,
                    help="clips per GPU of the configs[4] leg (8192 / 8 GPUs), run when more than one rank is launched")
    ap.add_argument("--config4-batch", type=int, default=64, help="A-GCN clips of the configs[3] clip leg")
    ap.add_argument("--config4-streams", type=int, default=1024, help="CoAGCN streams of the configs[3] online leg")
    ap.add_argument("--config4-shards", type=int, default=1,
                    help="stream shards of the CoAGCN leg (rounds 3-5 ran three; with the slot-balanced temporal-step tiles one shard "
                         "measures fastest: 1039 k frames/s against 1034 k / 967 k with two / three)")
    ap.add_argument("--no-split-leg", action="store_true", help="skip the opt-in bf16x3 precision-mode leg")
    ap.add_argument("--no-config5-leg", action="store_true", help="N = 1 only: skip the configs[4] per-GPU shard leg (1024 clips in one forward)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency-leg", action="store_true", help="skip the 1- and 16-stream per-frame latency leg of the online workload")
    ap.add_argument("--latency-frames", type=int, default=400, help="frames timed per stream count by the latency leg")
    ap.add_argument("--no-clip-latency-leg", action="store_true", help="skip the batch-1 / batch-8 clip latency leg (N = 1 only)")
    ap.add_argument("--clip-latency-forwards", type=int, default=200, help="forwards timed per batch size and mode by the clip latency leg")
    ap.add_argument("--cpu-budget", type=float, default=50.0, help="seconds of CPU work the configs[1]/[2] cpu_baseline legs may take in all")
    ap.add_argument("--cpu-budget-config4", type=float, default=14.0, help="seconds of CPU work for the two configs[3] cpu_baseline legs")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(args.gpus))            # parent: starts the ranks, relays rank 0's line, never touches a GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started as one rank of a WORLD_SIZE={world} job: the two must agree")
    import torch.distributed as dist
    # CSK_BENCH_BACKEND=gloo (debugging aid): lets several ranks share one GPU, which RCCL refuses -- used to exercise the
    # N > 1 code path of this script on a 1-GPU box; real runs use the default, "nccl" = RCCL, one rank per GPU
    backend = os.environ.get("CSK_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    if os.environ.get("CSK_BENCH_SAME_DEVICE") == "1":     # test hook: every rank on device 0 (what a launcher that drops LOCAL_RANK does)
        dev_index = 0
    # CSK_BENCH_FORCE_DIST=1: initialise the process group and run every collective of the N > 1 path with ONE rank as
    # well (what a 1-GPU box can exercise of RCCL: tests/test_gpu_multirank.py)
    use_dist = world > 1 or (os.environ.get("CSK_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    import _bootstrap
    pkg = _bootstrap.load()
    from continual_skeletons_amd import parallel

    do_clip, do_step = args.workload in ("both", "clip"), args.workload in ("both", "step")
    cpu = cpu_step = None
    if rank == 0 and not args.no_cpu_baseline:
        host_threads = host_cpu_threads()                   # one thread count for both legs
        if do_clip:
            cpu = cpu_baseline_clip(seed=0, threads=host_threads, budget_s=0.6 * args.cpu_budget)
        if do_step:
            cpu_step = cpu_baseline_step(seed=0, threads=host_threads, budget_s=0.4 * args.cpu_budget)

    def max_over_ranks(v):
        t = torch.tensor([v], device=dev, dtype=torch.float64)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def device_ident():
        import socket
        try:
            return f"{socket.gethostname()}/{torch.cuda.get_device_properties(dev).uuid}"
        except Exception:
            return f"{socket.gethostname()}/cuda:{dev_index}"

    def per_rank_table(local_ms_per_step):
        """[{rank, device_uuid, ms_per_step}] over all ranks (fixed-size byte tensors through all_gather: works with RCCL and
        gloo alike); `ranks_seen` = distinct devices among them, == world when every rank has a GPU of its own."""
        ident = device_ident().encode()[:96]
        if not use_dist:
            return [{"rank": 0, "device_uuid": ident.decode(), "ms_per_step": round(local_ms_per_step, 3)}]
        buf = torch.zeros(104, dtype=torch.uint8, device=dev)
        buf[:len(ident)] = torch.tensor(list(ident), dtype=torch.uint8, device=dev)
        ms = torch.tensor([local_ms_per_step], dtype=torch.float64, device=dev).view(torch.uint8)
        buf[96:104] = ms
        allb = [torch.zeros_like(buf) for _ in range(world)]
        dist.all_gather(allb, buf)
        rows = []
        for r, b in enumerate(allb):
            b = b.cpu()
            name = bytes(b[:96].tolist()).rstrip(b"\0").decode(errors="replace")
            rows.append({"rank": r, "device_uuid": name, "ms_per_step": round(float(b[96:104].clone().view(torch.float64).item()), 3)})
        return rows

    def ranks_seen(rows):
        return len({r["device_uuid"] for r in rows})

    line = None
    B = args.batch

    def clip_leg(batch, steps, warmup, seed0=100, precision="f32", info=None):
        """`steps` timed clip forwards of `batch` clips per rank (+ the logit all-gather when ranks > 1), bracketed by
        barrier + synchronize; returns (max-over-ranks seconds, tcn_stage HIP-event ms, launches timed).  precision
        "bf16x3": the opt-in split arithmetic of the temporal conv (blocks.set_precision); info["max_abs_diff_vs_f32"] then
        receives the largest logit difference against the default path on the same weights and input."""
        net = pkg.StGcn(pkg.ntu_graph().A, input_shape=(NTU["C"], NTU["T"], NTU["V"], NTU["M"]), num_classes=NTU["classes"]).eval()
        randomise_(net, seed=0)                              # identical weights on every rank
        net = net.to(dev)
        if precision != "f32":
            pkg.set_precision(net, precision)
        x = torch.rand((batch, NTU["C"], NTU["T"], NTU["V"], NTU["M"]), device=dev,
                       generator=torch.Generator(device=dev).manual_seed(seed0 + rank))

        def step():
            logits = net(x)
            return parallel.all_gather_logits(logits) if use_dist else logits

        with LaunchTimer(pkg, "tcn_stage") as lt:
            for _ in range(warmup):
                out = step()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            lt.enabled = True
            t0 = time.perf_counter()
            for _ in range(steps):
                out = step()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            lt.enabled = False
            tcn_ms = lt.total_ms()
            n_launch = len(lt.records)
        assert out.shape == (batch * world, NTU["classes"]) and bool(torch.isfinite(out).all())
        if info is not None:
            info["local_ms_per_step"] = dt / steps * 1e3           # this rank's own clock (per_rank table)
        dt = max_over_ranks(dt)
        if info is not None and precision != "f32":
            got = net(x)
            pkg.set_precision(net, "f32")
            ref = net(x)
            info["max_abs_diff_vs_f32"] = float((got - ref).abs().max())
            info["logit_absmax"] = float(ref.abs().max())
            del got, ref
        del x, out, net
        gc.collect()                 # engines hold reference cycles (state-dict hooks); free their slabs now
        torch.cuda.empty_cache()
        return dt, tcn_ms, n_launch

    per_rank = None
    if do_clip:
        inf1 = {}
        dt, tcn_ms, n_launch = clip_leg(B, args.steps, args.warmup, info=inf1)
        per_rank = per_rank_table(inf1["local_ms_per_step"])
        clips = B * world * args.steps
        flops_launch = tcn_flops_per_clip_forward(B * NTU["M"]) / 10.0       # average over the 10 launches / step
        avg_launch_s = tcn_ms / 1e3 / max(1, n_launch)
        achieved = flops_launch / avg_launch_s / 1e12
        traffic = load_traffic(batch=B)
        cfa, cfe, cby = workmodel.clip_totals(B * NTU["M"])     # per rank and step: FLOPs (SURVEY accounting), executed, bytes
        line = {
            "metric": "clips/sec (ST-GCN clip forward, NTU-60 shape; CoST-GCN online step reported under costgcn_online)",
            "value": round(clips / dt, 2),
            "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"ST-GCN 10-block clip forward, batch {B}/GPU, NTU-60 (3,300,25,2), fp32 [configs[1]]",
                       "global_batch": B * world, "frames_per_clip": NTU["T"], "parallelism": f"batch-shard x{world}",
                       "skeleton_frames_per_s": round(clips / dt * NTU["T"], 1)},
            "roofline": {"bound": "mfma", "kernel": "tcn_stage_kernel + tcn_stage16_kernel (the csk_tcn_stage_f32 launches)", "achieved": round(achieved, 2),
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 6),
                         "avg_launch_ms": round(avg_launch_s * 1e3, 4), "launches_timed": n_launch,
                         "flops_per_launch": flops_launch,
                         "traffic": traffic["hbm_bytes_per_launch"] if traffic else None,
                         "traffic_source": (f"{traffic.get('source')} (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes at this batch, "
                                            "not collected by this run)") if traffic else "no committed PMC pass at this batch size"},
            # whole-config roofline (SURVEY 8d / BASELINE.md 3): algorithmic FLOPs and fused-block minimum bytes of one
            # step (all ranks) against the time of one step; dense aggregation credited under flops_alg, only the
            # non-zeros the sparse kernel executes under flops_executed
            "roofline_config": workmodel.roofline_config(cfa * world, cby * world, dt / args.steps, cfe * world, n_gpus=world),
            "cpu_baseline": cpu,
            "ranks_seen": ranks_seen(per_rank), "per_rank": per_rank, "collective_backend": backend if use_dist else None,
        }
        if not args.no_split_leg:
            # OPT-IN precision mode "bf16x3" (csrc/tcn_split.hip): the same workload with the temporal conv on the bf16 matrix
            # pipe (fp32 operands split into 3 bf16 pieces, 6 piece products, fp32 accumulation).  NOT the headline and NOT
            # fp32: its own key, its own dtype, priced both ways.
            inf3 = {}
            dt3, tcn3_ms, n3 = clip_leg(B, args.steps, args.warmup, precision="bf16x3", info=inf3)
            tcn_fl = tcn_flops_per_clip_forward(B * NTU["M"])                # fp32-equivalent FLOPs of the ten temporal convs
            line["clip_bf16x3"] = {
                "metric": "clips/sec (ST-GCN clip forward, temporal conv in the opt-in bf16x3 split arithmetic)",
                "value": round(B * world * args.steps / dt3, 2), "unit": "clips/s", "ms_per_step": round(dt3 / args.steps * 1e3, 3),
                "dtype": "bf16x3-split, f32 accumulate (temporal conv; channel mix of the graph conv on 128-row layers); f32 (adjacency aggregation, 64-channel graph convs, head)",
                "speedup_vs_f32": round(dt / dt3, 3),
                "max_abs_logit_diff_vs_f32": inf3.get("max_abs_diff_vs_f32"), "logit_absmax": inf3.get("logit_absmax"),
                "tcn_stage": {"avg_launch_ms": round(tcn3_ms / max(1, n3), 4), "launches_timed": n3,
                              "fp32_equivalent_tflops": round(tcn_fl / 10.0 / (tcn3_ms / 1e3 / max(1, n3)) / 1e12, 2),
                              "executed_bf16_tflops": round(6 * tcn_fl / 10.0 / (tcn3_ms / 1e3 / max(1, n3)) / 1e12, 2),
                              "frac_of_bf16_peak": round(6 * tcn_fl / 10.0 / (tcn3_ms / 1e3 / max(1, n3)) / 1e12 / 2500.0, 4),
                              "note": "6 bf16 MFMA products per fp32 product: executed FLOPs = 6 x the fp32-equivalent; peak 2.5 PFLOP/s dense bf16"},
                "fp32_path_tcn_stage_avg_launch_ms": round(avg_launch_s * 1e3, 4),
                "scope": "clip kernels only (csk_tcn_stage_bf16x3, csk_gcn_stage_bf16x3); the continual step kernels stay exact fp32 (DESIGN.md)"}
        if world == 1 and not args.no_clip_latency_leg:
            line["clip_latency"] = run_clip_latency(pkg, dev, cpu, forwards=args.clip_latency_forwards)
        if use_dist or not args.no_config5_leg:
            # BASELINE.json configs[4]: batch 8192 over 8 GPUs = 1024 clips per GPU + the RCCL logit all-gather.  Reported
            # beside the 256 / GPU weak-scaling headline (which stays comparable with the N = 1 line); at N = 8 this IS
            # configs[4], at other N the same per-GPU shard (N = 1: one rank's shard, no collective).
            b5 = args.config5_batch
            steps5 = max(2, args.steps // 4)
            dt5, tcn5_ms, n5 = clip_leg(b5, steps5, 1, seed0=300)
            f5a, f5e, f5y = workmodel.clip_totals(b5 * NTU["M"])
            fl5 = tcn_flops_per_clip_forward(b5 * NTU["M"]) / 10.0
            ach5 = fl5 / (tcn5_ms / 1e3 / max(1, n5)) / 1e12
            traffic5 = load_traffic(batch=b5)
            line["config5"] = {
                "workload": f"ST-GCN clip inference, {b5} clips/GPU x {world} GPUs = global batch {b5 * world}, " +
                            (f"logit all-gather over {backend}" if use_dist else "one rank's shard, no collective") +
                            f" [configs[4]{'' if (b5 == 1024 and world == 8) else ' per-GPU shard shape'}]",
                "value": round(b5 * world * steps5 / dt5, 2), "unit": "clips/s", "clips_per_gpu": b5, "global_batch": b5 * world,
                "steps": steps5, "ms_per_step": round(dt5 / steps5 * 1e3, 3),
                # the dominant kernel of this leg, timed live on rank 0 exactly as the headline's (HIP events around every launch)
                "roofline": {"bound": "mfma", "kernel": "tcn_stage_kernel + tcn_stage16_kernel (the csk_tcn_stage_f32 launches)", "achieved": round(ach5, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                             "unit": "TFLOP/s", "frac": round(ach5 / PEAK_F32_MFMA_TFLOPS, 6),
                             "avg_launch_ms": round(tcn5_ms / max(1, n5), 4), "launches_timed": n5, "flops_per_launch": fl5,
                             "traffic": traffic5["hbm_bytes_per_launch"] if traffic5 else None,
                             "traffic_source": (f"{traffic5.get('source')} (committed rocprofv3 --pmc passes at this batch, not collected by this run)")
                             if traffic5 else "no committed PMC pass at this batch size"},
                "roofline_config": workmodel.roofline_config(f5a * world, f5y * world, dt5 / steps5, f5e * world, n_gpus=world)}

    if do_step:
        sdt, stcn_ms, sn, sbytes = run_step_workload(pkg, dev, args.streams, args.step_cycles, 2, rank, world, parallel, dist, args.stream_shards, fpl=args.frames_per_launch, use_dist=use_dist)
        sdt_local = sdt
        sdt = max_over_ranks(sdt)
        # kernel-level timing: launches must not overlap and must go through the Python hook -> short single-shard pass
        gc.collect()                 # engines hold reference cycles (state-dict hooks); free their slabs now
        torch.cuda.empty_cache()
        kcycles = max(1, args.kernel_cycles)
        _, stcn_ms, sn, _ = run_step_workload(pkg, dev, args.streams, kcycles, 1, rank, world, parallel, dist, 1, native_plan=False, use_dist=use_dist)
        # the timed launches are the tcn_step_kernel launches of a cycle: blocks 5-10 (blocks 1-4, C_out = 64, advance with
        # the fused csk_co_block_step_f32 launch, which does not go through the hook)
        step_layers = [l for l in workmodel.step_layers(4) if l["co"] > 64]
        tfl = 2.0 * args.streams * NTU["M"] * sum(l["tcn_macs"] for l in step_layers)
        assert sn == kcycles * len(step_layers), (sn, len(step_layers))
        fps = args.frames_per_launch * args.streams * world * args.step_cycles / sdt
        ach = tfl * kcycles / (stcn_ms / 1e3) / 1e12 if stcn_ms > 0 else 0.0
        straffic = load_traffic("traffic_tcn_step.json", streams=args.streams)
        sfa, sfe, sby = workmodel.step_totals(args.streams * NTU["M"], args.frames_per_launch)   # per rank and cycle
        thr = None
        if args.frames_per_launch != 8:      # throughput mode: two stride cycles per launch (adds 4 frames of latency)
            gc.collect()
            torch.cuda.empty_cache()
            tdt, _, _, _ = run_step_workload(pkg, dev, args.streams, args.step_cycles, 2, rank, world, parallel, dist,
                                             args.stream_shards, fpl=8, use_dist=use_dist)
            tdt = max_over_ranks(tdt)
            tfa, tfe, tby = workmodel.step_totals(args.streams * NTU["M"], 8)
            thr = {"frames_per_launch": 8, "value": round(8 * args.streams * world * args.step_cycles / tdt, 1),
                   "unit": "frames/s",
                   "roofline_config_frac": workmodel.roofline_config(tfa * world, tby * world, tdt / args.step_cycles, tfe * world, n_gpus=world)["frac"]}
        step_info = {"metric": "skeleton frames/sec (CoST-GCN online step, NTU-60 shape)", "value": round(fps, 1),
                     "unit": "frames/s", "streams_per_gpu": args.streams, "stream_shards": args.stream_shards, "frames_per_launch": args.frames_per_launch, "ms_per_frame_step": round(sdt / args.step_cycles / args.frames_per_launch * 1e3, 4),
                     "predictions_per_s": round(fps / 4, 1), "state_slab_GB_per_gpu": round(sbytes[0] / 1e9, 3),
                     "split_k_scratch_GB_per_gpu": round(sbytes[1] / 1e9, 3),
                     "roofline": {"bound": "mfma", "kernel": "tcn_step16_kernel", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                                  "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 6), "launches_timed": sn,
                                  "avg_launch_ms": round(stcn_ms / max(1, sn), 4),
                                  "flops_per_launch": tfl / len(step_layers),
                                  "timing": "single stream shard, launches driven from Python with HIP events around every "
                                            f"csk_tcn_step_f32 launch (tcn_step16_kernel; blocks 5-10 -- blocks 1-4 advance through the fused block / "
                                            f"stack entry points) of {kcycles} cycles of "
                                            "4 frames: the launch shape of tools/online_pass.py --shards 1, whose rocprofv3 per-layer "
                                            "table is profiles/r06_online_1shard.md (rows L5-L10 tcn_step)",
                                  "traffic": straffic["hbm_bytes_per_launch"] if straffic else None,
                                  "traffic_source": (f"{straffic.get('source')} (committed PMC passes, not collected by this run)")
                                  if straffic else "no committed PMC pass at this stream count"},
                     "roofline_config": workmodel.roofline_config(sfa * world, sby * world, sdt / args.step_cycles, sfe * world, n_gpus=world),
                     "throughput_mode": thr, "cpu_baseline": cpu_step, "config": "BASELINE.json configs[2]"}
        if world == 1 and not args.no_latency_leg:
            gc.collect()
            torch.cuda.empty_cache()
            step_info["latency"] = run_latency(pkg, dev, cpu_step, frames=args.latency_frames)
        if line is None:          # --workload step: the online metric is the primary one
            line = {"metric": step_info["metric"], "value": step_info["value"], "unit": "frames/s", "n_gpus": world,
                    "steps": args.step_cycles, "warmup": 2, "ms_per_step": round(sdt / args.step_cycles * 1e3, 3),
                    "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                    "config": {"workload": f"CoST-GCN online step, {args.streams} streams/GPU, NTU-60, one step = {args.frames_per_launch} frames [configs[2]]",
                               "parallelism": f"stream-shard x{world}"},
                    "roofline": step_info["roofline"], "roofline_config": step_info["roofline_config"], "cpu_baseline": cpu_step,
                    "latency": step_info.get("latency")}
        else:
            line["costgcn_online"] = step_info
    if do_clip and do_step and world == 1:          # BASELINE.json configs[3] beside the headline numbers (per GPU)
        gc.collect()
        torch.cuda.empty_cache()
        line["agcn_kinetics"] = run_config4(pkg, dev, parallel, batch=args.config4_batch, streams=args.config4_streams,
                                            shards=min(args.config4_shards, max(1, args.config4_streams)), cpu_threads=0 if args.no_cpu_baseline else host_cpu_threads(),
                                            cpu_budget=args.cpu_budget_config4)
    if per_rank is None:                                 # --workload step: the table carries the online leg's clock
        per_rank = per_rank_table(sdt_local / args.step_cycles * 1e3)
    line.setdefault("per_rank", per_rank)
    line.setdefault("ranks_seen", ranks_seen(per_rank))
    line.setdefault("collective_backend", backend if use_dist else None)
    # one rank per GPU is the contract of an RCCL run: ranks that share a device (a launcher that did not pin LOCAL_RANK)
    # would print a throughput that is not N GPUs' -- the line is still printed, marked invalid, and the exit code says so
    bad = use_dist and backend == "nccl" and line["ranks_seen"] != world
    if bad:
        line["error"] = f"ranks_seen {line['ranks_seen']} != n_gpus {world}: ranks share a device; this line is INVALID"
    if rank == 0:
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()
    if bad:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
