#!/usr/bin/env bash
# PMC passes of the online path (1 shard): SQ wave/wait/MFMA counters and LDS counters per kernel instantiation
set -uo pipefail
tag="${1:-r06p}"
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$R/gpurun_out/prof_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32"
rocprofv3 --kernel-trace --pmc $A --output-format csv -d "$out/sqa" -- python3 "$R/tools/online_pass.py" --shards 1 --cycles 8 --warm-cycles 60 > "$out/sqa.log" 2>&1
rocprofv3 --kernel-trace --pmc $B --output-format csv -d "$out/sqb" -- python3 "$R/tools/online_pass.py" --shards 1 --cycles 8 --warm-cycles 60 > "$out/sqb.log" 2>&1
tail -2 "$out/sqb.log"
cd "$R"
python3 - "$out" <<'PY'
import csv, glob, collections, sys, os
out = sys.argv[1]
def short(n):
    for k in ("co_stack16_kernel", "tcn_step16_kernel", "gcn16_kernel", "tcn_step_kernel", "gcn_stage_sparse2_kernel", "co_head_kernel"):
        if k in n:
            return k + (n[n.find("<"): n.find(">") + 1] if "<" in n else "")
    return None
for sub in ("sqa", "sqb"):
    files = sorted(glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    if not files:
        print("no counters for", sub); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(files[-1])):
        k = short(r["Kernel_Name"])
        if not k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    names = sorted({c for k in acc for c in acc[k]})
    print("## " + sub); print("| kernel | n | avg us | " + " | ".join(names) + " |")
    for k in sorted(acc, key=lambda kk: -sum(dur[kk].values())):
        n = len(dur[k]); print(f"| {k} | {n} | {sum(dur[k].values()) / n:.1f} | " + " | ".join(f"{sum(acc[k][c]) / max(len(acc[k][c]), 1):.4g}" for c in names) + " |")
PY
find "$out" -name "*agent_info.csv" -delete; find "$out" -name "*kernel_trace.csv" -delete; find "$out" -name "*counter_collection.csv" -delete
