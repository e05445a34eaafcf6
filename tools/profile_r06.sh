#!/usr/bin/env bash
# Run ON THE GPU BOX (via gpurun): every rocprofv3 pass behind the round-6 profile summaries.
#   kernel traces: clip forward (exact fp32 and the opt-in bf16x3 mode), online path (1 and 2 stream shards), config 4
#               few-stream latency path (1 and 16 streams, one frame per call, latency mode)
#   PMC passes (separate runs, --kernel-trace only beside --pmc): MFMA busy / wave cycles and FETCH / WRITE bytes of the
#   clip forward in both precision modes, of the A-GCN clip forward, and of the ONLINE shapes (tools/online_pass.py
#   --shards 1, CoST-GCN and CoAGCN: tcn_step / co_block / step-shape GCN / attention launches)
#   round 6: the online passes run the slot-balanced tiles of csrc/step16.hip (one stream shard is the default; the two-shard
#   trace stays for comparison) and an LDS-counter pass of the CoST-GCN online shapes is added
#   (round 5:) FETCH / WRITE passes of the batch-1024 clip forward (bench.py's config5 leg), kernel trace of the batch-1
#   clip forward in latency mode (tools/clip_latency_pass.py)
# usage: bash tools/profile_r05.sh <tag>
set -uo pipefail
tag="${1:-r06}"
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$R/gpurun_out/prof_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for prec in f32 bf16x3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/clip_$prec" -- python3 "$R/tools/clip_pass.py" --precision $prec > "$out/clip_$prec.log" 2>&1
  grep CLIP_PASS "$out/clip_$prec.log"
done
for sh in 1 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/online$sh" -- python3 "$R/tools/online_pass.py" --shards $sh > "$out/online$sh.log" 2>&1
  grep ONLINE_PASS "$out/online$sh.log"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/agcn_clip" -- python3 "$R/tools/agcn_prof.py" 64 6 > "$out/agcn_clip.log" 2>&1
grep AGCN_PASS "$out/agcn_clip.log"
for sh in 1 2 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/coagcn$sh" -- python3 "$R/tools/online_pass.py" --model coagcn --shards $sh --cycles 16 > "$out/coagcn$sh.log" 2>&1
  grep ONLINE_PASS "$out/coagcn$sh.log"
done
for st in 1 16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/latency$st" -- python3 "$R/tools/latency_pass.py" --streams $st > "$out/latency$st.log" 2>&1
  grep "ms per frame" "$out/latency$st.log"
done
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
for prec in f32 bf16x3; do
  rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d "$out/sq_$prec" -- python3 "$R/tools/clip_pass.py" --precision $prec --forwards 2 > "$out/sq_$prec.log" 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch_$prec" -- python3 "$R/tools/clip_pass.py" --precision $prec --forwards 2 > "$out/fetch_$prec.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write_$prec" -- python3 "$R/tools/clip_pass.py" --precision $prec --forwards 2 > "$out/write_$prec.log" 2>&1
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch_f32_b1024" -- python3 "$R/tools/clip_pass.py" --precision f32 --batch 1024 --forwards 1 > "$out/fetch_f32_b1024.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write_f32_b1024" -- python3 "$R/tools/clip_pass.py" --precision f32 --batch 1024 --forwards 1 > "$out/write_f32_b1024.log" 2>&1
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d "$out/sq_f32_b1024" -- python3 "$R/tools/clip_pass.py" --precision f32 --batch 1024 --forwards 1 > "$out/sq_f32_b1024.log" 2>&1
for b in 1 8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/clip_latency_b$b" -- python3 "$R/tools/clip_latency_pass.py" --batch $b --split-k 4 --trace > "$out/clip_latency_b$b.log" 2>&1
done
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d "$out/sq_agcn" -- python3 "$R/tools/agcn_prof.py" 64 3 > "$out/sq_agcn.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch_agcn" -- python3 "$R/tools/agcn_prof.py" 64 3 > "$out/fetch_agcn.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write_agcn" -- python3 "$R/tools/agcn_prof.py" 64 3 > "$out/write_agcn.log" 2>&1
for model in costgcn coagcn; do
  rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d "$out/sq_online_$model" -- python3 "$R/tools/online_pass.py" --model $model --shards 1 --cycles 8 --warm-cycles 60 > "$out/sq_online_$model.log" 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch_online_$model" -- python3 "$R/tools/online_pass.py" --model $model --shards 1 --cycles 8 --warm-cycles 60 > "$out/fetch_online_$model.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write_online_$model" -- python3 "$R/tools/online_pass.py" --model $model --shards 1 --cycles 8 --warm-cycles 60 > "$out/write_online_$model.log" 2>&1
done
LDSC="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"
rocprofv3 --kernel-trace --pmc $LDSC --output-format csv -d "$out/lds_online_costgcn" -- python3 "$R/tools/online_pass.py" --shards 1 --cycles 8 --warm-cycles 60 > "$out/lds_online_costgcn.log" 2>&1
python3 - "$out" <<'PY' > "$out/lds_online_costgcn.md"
import csv, glob, collections, sys, os
out = sys.argv[1]
def short(n):
    for k in ("co_stack16_kernel", "tcn_step16_kernel", "gcn16_kernel", "co_head_kernel"):
        if k in n:
            return k + (n[n.find("<"): n.find(">") + 1] if "<" in n else "")
    return None
files = sorted(glob.glob(f"{out}/lds_online_costgcn/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(dict)
for r in csv.DictReader(open(files[-1])):
    k = short(r["Kernel_Name"])
    if not k: continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("| kernel | launches | avg us | LDS busy (LDS_IDX_ACTIVE / (8 x SQ_BUSY_CYCLES)) | bank-conflict share of LDS cycles | LDS instr per wave-cycle x1000 |")
print("|---|---|---|---|---|---|")
for k in sorted(acc, key=lambda kk: -sum(dur[kk].values())):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    n = len(dur[k])
    print(f"| {k} | {n} | {sum(dur[k].values()) / n:.1f} | {c['SQ_LDS_IDX_ACTIVE'] / (8 * c['SQ_BUSY_CYCLES']):.3f} | {c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1):.3f} | {c['SQ_INSTS_LDS']:.3g} |")
PY
find "$out" -name "*kernel_trace.csv" -path "*lds_*" -delete; find "$out" -name "*counter_collection.csv" -path "*lds_*" -delete
find "$out" -name "*agent_info.csv" -delete
find "$out" -name "*kernel_trace.csv" -path "*sq_*" -delete; find "$out" -name "*kernel_trace.csv" -path "*fetch_*" -delete; find "$out" -name "*kernel_trace.csv" -path "*write_*" -delete
du -sh "$out"
