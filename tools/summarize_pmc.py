#!/usr/bin/env python3
"""Condense the PMC passes of tools/profile_r06.sh (gpurun_out/prof_<tag>/{sq,fetch,write}_<mode>) into
profiles/<out>_pmc_summary.md: per kernel instantiation and precision mode -- launches, clock, MFMA-pipe busy fraction,
waves per SIMD, wait fractions, HBM read / write MB per launch (FETCH_SIZE calibrated on input_norm_kernel, whose bytes
are known: MI355X_MICROARCH.md HBM section) -- and refresh profiles/traffic_tcn_stage.json and traffic_tcn_step.json (what
bench.py prints as roofline.traffic for the clip / online legs) from the exact-fp32 clip pass and the CoST-GCN online pass.
usage: python tools/summarize_pmc.py <tag> [out_tag]"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows_of(pattern):
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return list(csv.DictReader(open(files[-1]))) if files else []


def short(name):
    for k in ("co_stack16_kernel", "tcn_step16_kernel", "tcn_stage16_kernel", "gcn16_kernel", "tcn_split_stage_kernel", "gcn_split_stage_kernel", "tcn_stage_kernel", "gcn_stage_sparse2_kernel", "gcn_stage_dense2_kernel", "gcn_stage_dense_kernel", "gcn_stage_kernel",
              "agcn_embed_attention_kernel", "agcn_softmax_parts_kernel", "tcn_step_kernel", "co_block_kernel", "step_reduce_kernel", "co_head_kernel",
              "agcn_logits_partial_kernel", "agcn_softmax_kernel", "input_norm_kernel", "pool_kernel", "fc_kernel"):
        if k in name:
            t = name[name.find("<"): name.find(">") + 1] if "<" in name else ""
            return k + t
    return None


def pmc(d, sub):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    seen = set()
    for r in rows_of(f"{d}/{sub}/**/*counter_collection.csv"):
        k = short(r["Kernel_Name"])
        if k is None:
            continue
        out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        key = (r["Dispatch_Id"], k)
        if key not in seen:
            seen.add(key)
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    return out, dur


def main(tag, out_tag):
    d = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    L = [f"# PMC passes ({out_tag}): tools/profile_r06.sh -- `rocprofv3 --kernel-trace --pmc ...` of tools/clip_pass.py (batch 256, both precision "
         "modes), tools/agcn_prof.py (A-GCN, Kinetics shape, batch 64) and tools/online_pass.py --shards 1 (the online shapes: CoST-GCN and "
         "CoAGCN, 1024 streams, 4-frame cycles); separate passes for SQ counters, FETCH_SIZE and WRITE_SIZE", ""]
    stage_entries = []
    for mode, title in (("f32", "ST-GCN clip forward, exact fp32"), ("f32_b1024", "ST-GCN clip forward, exact fp32, batch 1024 (one rank's shard of configs[4])"),
                        ("bf16x3", "ST-GCN clip forward, opt-in bf16x3 temporal conv"),
                        ("agcn", "A-GCN clip forward (config 4)"),
                        ("online_costgcn", "online: CoST-GCN, 1024 streams, one stream shard, 4 frames per launch (configs[2] launch shapes)"),
                        ("online_coagcn", "online: CoAGCN, Kinetics shape, 1024 streams, one stream shard, 4 frames per launch (configs[3])")):
        sq, dur = pmc(d, f"sq_{mode}")
        fetch, _ = pmc(d, f"fetch_{mode}")
        write, _ = pmc(d, f"write_{mode}")
        if not sq:
            continue
        batch = 1024 if mode.endswith("b1024") else 256
        known = batch * 3 * 300 * 25 * 2 * 4
        cal_raw = fetch.get("input_norm_kernel", {}).get("FETCH_SIZE", [])
        cal = known / (max(cal_raw) * 1024) if cal_raw and mode in ("f32", "bf16x3", "f32_b1024") else 2.0
        L += [f"## {title}", "", f"(FETCH_SIZE calibration on input_norm_kernel, {known / 1e6:.2f} MB known: factor {cal:.3f})" if mode in ("f32", "bf16x3", "f32_b1024") else
              "(FETCH_SIZE x 2: the gfx950 correction of MI355X_MICROARCH.md, as calibrated in the ST-GCN passes)", "",
              "| kernel | launches | avg ms | clock GHz | MFMA busy | waves/SIMD | WAIT_ANY/wave | WAIT_INST/wave | HBM read MB | HBM write MB |",
              "|---|---|---|---|---|---|---|---|---|---|"]
        traffic = {}
        for k in sorted(sq, key=lambda kk: -sum(dur[kk])):
            c = sq[k]
            g = sum(c.get("GRBM_GUI_ACTIVE", [0])) / 8
            if g == 0 or ("stage" not in k and "agcn" not in k and not (mode.startswith("online") and ("step" in k or "co_" in k or "gcn16" in k))):
                continue
            wc = sum(c["SQ_WAVE_CYCLES"])
            secs = sum(dur[k]) / 1e3
            f = fetch.get(k, {}).get("FETCH_SIZE", [])
            w = write.get(k, {}).get("WRITE_SIZE", [])
            rd = (sum(f) / len(f)) * 1024 * cal / 1e6 if f else float("nan")
            wr = (sum(w) / len(w)) * 1024 / 1e6 if w else float("nan")
            traffic[k] = dict(launches=len(dur[k]), read_MB=rd, write_MB=wr)
            L.append(f"| {k} | {len(dur[k])} | {sum(dur[k]) / len(dur[k]):.4f} | {g / secs / 1e9:.2f} | {sum(c['SQ_VALU_MFMA_BUSY_CYCLES']) / (g * 1024):.3f} | "
                     f"{wc * 4 / (g * 1024):.2f} | {sum(c['SQ_WAIT_ANY']) / wc:.3f} | {sum(c['SQ_WAIT_INST_ANY']) / wc:.3f} | {rd:.1f} | {wr:.1f} |")
        L.append("")
        if mode == "online_costgcn":
            dom = [v for k, v in traffic.items() if k.startswith("tcn_step16_kernel") or k.startswith("tcn_step_kernel")]
            red = traffic.get("step_reduce_kernel")
            if dom and all(v["read_MB"] == v["read_MB"] for v in dom):
                n = sum(v["launches"] for v in dom)
                tot = sum((v["read_MB"] + v["write_MB"]) * 1e6 * v["launches"] for v in dom)
                if red and red["read_MB"] == red["read_MB"]:
                    tot += (red["read_MB"] + red["write_MB"]) * 1e6 * red["launches"]      # the split-K launches' reduction
                json.dump({"kernel": "tcn_step16_kernel", "hbm_bytes_per_launch": tot / n, "launches": n, "source": f"profiles/{out_tag}_pmc_summary.md",
                           "streams": 1024, "stream_shards": 1, "frames_per_launch": 4,
                           "note": "average over the stand-alone temporal-step launches of a cycle (block 1 and blocks 5-10; blocks 2-4 run inside co_stack16_kernel)"},
                          open(os.path.join(ROOT, "profiles", "traffic_tcn_step.json"), "w"), indent=1)
        if mode in ("f32", "f32_b1024"):
            dom = [v for k, v in traffic.items() if k.startswith("tcn_stage_kernel") or k.startswith("tcn_stage16_kernel")]
            if dom and all(v["read_MB"] == v["read_MB"] for v in dom):
                n = sum(v["launches"] for v in dom)
                hb = sum((v["read_MB"] + v["write_MB"]) * 1e6 * v["launches"] for v in dom) / n
                stage_entries.append({"kernel": "tcn_stage_kernel + tcn_stage16_kernel", "hbm_bytes_per_launch": hb, "launches": n,
                                      "source": f"profiles/{out_tag}_pmc_summary.md", "batch": batch})
                # one entry per profiled batch size (bench.py: load_traffic(batch=...)); the batch-256 entry stays at the top level too
                top = next((e for e in stage_entries if e["batch"] == 256), stage_entries[0])
                json.dump(dict(top, by_batch=stage_entries), open(os.path.join(ROOT, "profiles", "traffic_tcn_stage.json"), "w"), indent=1)
    open(os.path.join(ROOT, "profiles", f"{out_tag}_pmc_summary.md"), "w").write("\n".join(L) + "\n")
    print("\n".join(L))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])
