#!/usr/bin/env bash
# Run ON THE GPU BOX (via gpurun): one SQ counter pass over the A-GCN clip forward (tools/agcn_prof.py 64 3): MFMA / VALU
# busy, LDS bank conflicts, wait fractions per kernel.  Prints a per-kernel table.  usage: bash tools/profile_config4_pmc.sh <tag>
set -uo pipefail
tag="${1:-r03}"
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$R/gpurun_out/prof_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d "$out/agcn_sq" -- python3 "$R/tools/agcn_prof.py" 64 3 > "$out/agcn_sq.log" 2>&1
SQ2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS"
rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d "$out/agcn_sq2" -- python3 "$R/tools/agcn_prof.py" 64 3 > "$out/agcn_sq2.log" 2>&1
tail -2 "$out/agcn_sq2.log"
find "$out" -name "*agent_info.csv" -delete
python3 - "$out" <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
for sub in ("agcn_sq", "agcn_sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            k = n[:n.find("(")] if "(" in n else n
            k = k.replace("void ", "")[:60]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = sorted({c for k in acc for c in acc[k]})
    print("##", sub)
    print("| kernel | n | " + " | ".join(names) + " |")
    for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CYCLES", kv[1].get("SQ_INSTS_VALU", [0])))):
        if "gcn" not in k and "tcn" not in k:
            continue
        n = len(next(iter(d.values())))
        print(f"| {k} | {n} | " + " | ".join(f"{sum(d[c]) / max(len(d[c]), 1):.4g}" for c in names) + " |")
PY
