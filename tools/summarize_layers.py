#!/usr/bin/env python3
"""Per-kernel and PER-LAYER summary of a `rocprofv3 --kernel-trace` run of tools/online_pass.py (the online shape of
bench.py: CoST-GCN, 1024 streams, 4-frame cycles, native plan), written to profiles/<tag>.{md,csv}.

Only steady-state cycles are counted: per HIP stream the kernel sequence is cut into cycles at each run of
input_norm_kernel launches and the last --cycles cycles are kept (all ten blocks emit, pool window full).  Inside a
cycle the block kernels come in layer order: gcn_stage* then tcn_step* per block (or one co_block* kernel per block).
FLOPs per launch: tools/workmodel.py (SURVEY 8d accounting; dense aggregation credited under `alg`, non-zeros only
under `exec`).
With --mode clip the same is done for a trace of `bench.py --workload clip` (one cycle = one clip forward of --batch
clips: input_norm, then gcn_stage* + tcn_stage* per block).
usage: python tools/summarize_layers.py <trace_dir> <tag> [--mode online|clip] [--cycles 24] [--streams 1024] [--shards 1] [--fpl 4] [--batch 256]"""
import argparse
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import workmodel as wm  # noqa: E402


def short(name):
    for k in ("co_stack16_kernel", "tcn_step16_kernel", "gcn16_kernel", "gcn_stage_sparse2_kernel", "gcn_stage_sparse_kernel", "gcn_stage_kernel", "tcn_stage_kernel", "tcn_step_kernel", "pool_kernel", "co_block_kernel", "input_norm_kernel",
              "input_norm_frames_kernel", "co_head_kernel", "gcn_reduce_kernel", "co_spatial_pool_kernel", "co_window_mean_kernel", "fc_kernel", "step_reduce_kernel", "agcn_attention_step_kernel",
              "agcn_embed_attention_kernel", "agcn_softmax_parts_kernel", "agcn_attention_kernel", "agcn_logits_partial_kernel", "agcn_softmax_kernel", "tcn_split_stage_kernel", "gcn_split_stage_kernel", "gcn_stage_dense_kernel", "gcn_stage_dense2_kernel"):
        if k in name:
            t = name[name.find("<"): name.find(">") + 1] if "<" in name else ""
            return k + t
    return name[:48]


def klass(name):
    if "agcn_" in name:
        return "a"
    if "gcn_stage" in name or "gcn_split" in name or "gcn16" in name:
        return "g"
    if "tcn_step" in name or "tcn_stage" in name or "tcn_split" in name:
        return "t"
    if "co_stack" in name:
        return "s"
    if "co_block" in name:
        return "f"
    if "input_norm" in name:
        return "i"
    return "o"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace_dir")
    ap.add_argument("tag")
    ap.add_argument("--cycles", type=int, default=24)
    ap.add_argument("--streams", type=int, default=1024)
    ap.add_argument("--shards", type=int, default=1)
    ap.add_argument("--fpl", type=int, default=4)
    ap.add_argument("--note", default="")
    ap.add_argument("--stack-blocks", type=int, default=3, help="blocks one co_stack16_kernel launch covers (CoST-GCN: layers 2-4)")
    ap.add_argument("--mode", default="online", choices=["online", "clip"])
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--model", default="stgcn", choices=["stgcn", "agcn"],
                    help="agcn: A-GCN / CoAGCN at the Kinetics shape (V = 18): per block embedding conv (a tcn kernel with "
                         "k = 1 in front of the graph conv), attention, general graph conv, temporal conv")
    a = ap.parse_args()
    V = 18 if a.model == "agcn" else 25
    adaptive = a.model == "agcn"
    tcn_per_cycle = 10
    files = sorted(glob.glob(os.path.join(a.trace_dir, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    if not files:
        raise SystemExit(f"no kernel_trace.csv under {a.trace_dir}")
    rows = list(csv.DictReader(open(files[-1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    by_stream = collections.defaultdict(list)
    for r in rows:
        by_stream[r["Stream_Id"]].append(r)

    clip = a.mode == "clip"
    if clip:
        n_skel = a.batch * 2
        layers = [dict(l, frames_in=l["t_in"], emissions=l["t_out"]) for l in wm.clip_layers(V=V, adaptive=adaptive)]
    else:
        n_skel = a.streams // a.shards * 2
        layers = wm.step_layers(a.fpl, V=V, adaptive=adaptive)
    per_layer = [collections.defaultdict(list) for _ in range(10)]          # stage -> [ms]
    per_kernel = collections.defaultdict(list)
    cyc_kernel_ms, windows, other_ms = [], [], []
    used_streams = 0
    for sid, rs in by_stream.items():
        # cut into cycles at runs of input_norm launches
        cycles, cur, prev_i = [], None, False
        for r in rs:
            k = klass(r["Kernel_Name"])
            if k == "i" and not prev_i:
                cur = []
                cycles.append(cur)
            prev_i = k == "i"
            if cur is not None:
                cur.append(r)
        good = [c for c in cycles if sum((klass(r["Kernel_Name"]) in "tf") + a.stack_blocks * (klass(r["Kernel_Name"]) == "s") for r in c) == tcn_per_cycle]
        if len(good) < a.cycles:
            continue
        used_streams += 1
        good = good[-a.cycles:]
        # window of the kept cycles: from the first input_norm to the end of the last PATH kernel of the last cycle (what
        # the workload script launches after its timed loop -- isfinite checks, lazily loaded torch kernels -- is not part)
        path = [r for r in good[-1] if klass(r["Kernel_Name"]) != "o" or any(
            k in r["Kernel_Name"] for k in ("co_spatial_pool", "co_window_mean", "fc_kernel", "pool_kernel", "co_head"))]
        windows.append((int(good[0][0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in path)))
        for c in good:
            li, tot, oth, seen_a = 0, 0.0, 0.0, False
            for r in c:
                ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                k = klass(r["Kernel_Name"])
                tot += ms
                per_kernel[short(r["Kernel_Name"])].append(ms)
                if adaptive and li < 10 and (k == "a" or (k == "g" and not seen_a)):
                    # A-GCN block: embedding conv (the plain-mode graph-conv kernel in front of the attention), attention
                    # kernels (their durations are summed per block), graph conv, temporal conv
                    if k == "a":
                        if seen_a:
                            per_layer[li]["a"][-1] += ms
                        else:
                            per_layer[li]["a"].append(ms)
                        seen_a = True
                    else:
                        per_layer[li]["e"].append(ms)
                    continue
                if k in "tfs":
                    seen_a = False
                if k in "gtfs" and li < 10:
                    per_layer[li][k].append(ms)
                    if k in "tf":
                        li += 1
                    elif k == "s":
                        li += a.stack_blocks
                else:
                    oth += ms
            cyc_kernel_ms.append(tot)
            other_ms.append(oth)
    if not used_streams:
        raise SystemExit("no stream with enough steady-state cycles found")
    w0, w1 = min(w[0] for w in windows), max(w[1] for w in windows)
    wall_ms = (w1 - w0) / 1e6 / a.cycles
    fa, fe, by = (wm.clip_totals(a.batch * 2, V=V, adaptive=adaptive) if clip else
                  wm.step_totals(a.streams * 2, a.fpl, V=V, adaptive=adaptive))
    peak = wm.PEAK_F32_MFMA_TFLOPS

    if clip:
        L = [f"# Clip path, rocprofv3 --kernel-trace ({a.tag}): " + (f"tools/agcn_prof.py {a.batch} (A-GCN, Kinetics shape)" if adaptive else
             f"bench.py --workload clip --batch {a.batch}") + f"; last {a.cycles} forwards", ""]
    else:
        L = [f"# Online path, rocprofv3 --kernel-trace ({a.tag}): tools/online_pass.py " + ("--model coagcn " if adaptive else "") + f"--shards {a.shards} --fpl {a.fpl} "
             f"--streams {a.streams}; last {a.cycles} steady-state cycles of {used_streams} HIP stream(s)", ""]
    if a.note:
        L += [a.note, ""]
    L += ["## Per kernel instantiation (steady-state launches only)", "",
          "| kernel | launches | avg ms | min ms | max ms | total ms |", "|---|---|---|---|---|---|"]
    for k, v in sorted(per_kernel.items(), key=lambda kv: -sum(kv[1])):
        L.append(f"| {k} | {len(v)} | {sum(v) / len(v):.4f} | {min(v):.4f} | {max(v):.4f} | {sum(v):.3f} |")
    L += ["", (f"## Per layer (one launch covers {n_skel} skeleton sequences = batch {a.batch} x M=2)" if clip else
           f"## Per layer (one launch covers {n_skel} skeletons = {a.streams // a.shards} streams x M=2; a cycle = {a.fpl} frames)"), "",
          "| layer | stage | frames/emissions per launch | launches | avg ms | GFLOP alg (exec) per launch | TFLOP/s alg | frac of 157.3 | TFLOP/s exec | frac exec |",
          "|---|---|---|---|---|---|---|---|---|---|"]
    csv_rows = []
    sums = dict(g=[0.0, 0.0], t=[0.0, 0.0])
    for i, (pl, lw) in enumerate(zip(per_layer, layers)):
        for k in "eagtfs":
            if not pl[k]:
                continue
            avg = sum(pl[k]) / len(pl[k])
            if k == "s":           # one launch for layers i .. i + stack_blocks - 1 (graph conv + temporal step of each)
                grp = layers[i:i + a.stack_blocks]
                alg = sum(l["gcn_macs"] + l["agg_dense"] + l["tcn_macs"] for l in grp)
                ex = sum(l["gcn_macs"] + l["agg_sparse"] + l["tcn_macs"] for l in grp)
                cnt = lw["emissions"]
            elif k == "e":
                alg = ex = lw["embed_macs"]
                cnt = lw["frames_in"]
            elif k == "a":
                # no separate embedding launch in this block: the fused kernel (agcn_embed_attention_kernel) did both
                alg = ex = lw["attn_macs"] + (0 if pl["e"] else lw["embed_macs"])
                cnt = lw["frames_in"]
            elif k == "g":
                base = lw["gcn_macs"] - lw.get("embed_macs", 0) - lw.get("attn_macs", 0)
                alg, ex, cnt = base + lw["agg_dense"], base + lw["agg_sparse"], lw["frames_in"]
            elif k == "t":
                alg = ex = lw["tcn_macs"]
                cnt = lw["emissions"]
            else:
                alg, ex = lw["gcn_macs"] + lw["agg_dense"] + lw["tcn_macs"], lw["gcn_macs"] + lw["agg_sparse"] + lw["tcn_macs"]
                cnt = lw["emissions"]
            alg, ex = 2e-9 * alg * n_skel, 2e-9 * ex * n_skel
            tf = alg / avg
            name = {"e": "embed 1x1", "a": "attention" if pl["e"] else "embed + attention", "g": "gcn", "t": "tcn_stage" if clip else "tcn_step", "f": "fused",
                    "s": f"fused stack L{i + 1}-L{i + a.stack_blocks}"}[k]
            L.append(f"| L{i + 1} {lw['ci']}->{lw['co']} s{lw['stride']} | {name} | {cnt} | {len(pl[k])} | {avg:.4f} | {alg:.2f} ({ex:.2f}) | {tf:.1f} | {tf / peak:.3f} | {ex / avg:.1f} | {ex / avg / peak:.3f} |")
            csv_rows.append(dict(layer=i + 1, c_in=lw["ci"], c_out=lw["co"], stride=lw["stride"], stage=name, launches=len(pl[k]),
                                 avg_ms=round(avg, 5), gflop_alg=round(alg, 3), gflop_exec=round(ex, 3), tflops_alg=round(tf, 2),
                                 frac=round(tf / peak, 4), tflops_exec=round(ex / avg, 2), frac_exec=round(ex / avg / peak, 4)))
            kk = "g" if k in "eag" else "t"
            sums[kk][0] += avg
            sums[kk][1] += alg
    ksum = sum(cyc_kernel_ms) / len(cyc_kernel_ms)
    L += ["", "## Whole cycle", "",
          f"* kernel time per cycle and stream (sum of durations): {ksum:.4f} ms, of which non-block kernels (input norm, head, copies) {sum(other_ms) / len(other_ms):.4f} ms",
          f"* GCN-stage launches (embedding + attention + graph conv for A-GCN): {sums['g'][0]:.4f} ms for {sums['g'][1]:.1f} GFLOP -> {sums['g'][1] / max(sums['g'][0], 1e-9):.1f} TFLOP/s ({sums['g'][1] / max(sums['g'][0], 1e-9) / peak:.3f})"
          if sums["g"][0] else "* no separate GCN-stage launches",
          f"* TCN-step / fused launches: {sums['t'][0]:.4f} ms for {sums['t'][1]:.1f} GFLOP -> {sums['t'][1] / sums['t'][0]:.1f} TFLOP/s ({sums['t'][1] / sums['t'][0] / peak:.3f})",
          f"* wall time per cycle over all streams (first start to last end of the window / {a.cycles}): {wall_ms:.4f} ms "
          + (f"-> {a.batch / wall_ms * 1e3:,.0f} clips/s" if clip else f"-> {a.fpl * a.streams / wall_ms * 1e3:,.0f} frames/s"),
          f"* whole-config roofline: flops_alg {fa / 1e9:.1f} GFLOP (executed {fe / 1e9:.1f}), bytes_alg {by / 1e9:.3f} GB per cycle; "
          f"t_MFMA {fa / peak / 1e9:.4f} ms, t_HBM {by / wm.PEAK_HBM_TBS / 1e9:.4f} ms; "
          f"frac = t_roof / wall = {fa / peak / 1e9 / wall_ms:.3f} (executed FLOPs only: {fe / peak / 1e9 / wall_ms:.3f})"]
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    open(os.path.join(ROOT, "profiles", f"{a.tag}.md"), "w").write("\n".join(L) + "\n")
    with open(os.path.join(ROOT, "profiles", f"{a.tag}.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(csv_rows[0].keys()))
        w.writeheader()
        w.writerows(csv_rows)
    print("\n".join(L))


if __name__ == "__main__":
    main()
