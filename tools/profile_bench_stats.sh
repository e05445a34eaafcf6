#!/usr/bin/env bash
# Run ON THE GPU BOX (via gpurun): `rocprofv3 --kernel-trace --stats` of the DEFAULT bench command (python3 bench.py, CPU baseline
# legs skipped) -> profiles-ready markdown gpurun_out/prof_<tag>/bench_stats.md: rocprofv3's own per-kernel summary next to the
# JSON line the same run printed (whose roofline.avg_launch_ms must agree with the tcn_stage_kernel rows).
# usage: bash tools/profile_bench_stats.sh <tag>
set -uo pipefail
tag="${1:-r06}"
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$R/gpurun_out/prof_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/bench" -- python3 "$R/bench.py" --no-cpu-baseline > "$out/bench.log" 2> "$out/bench.err"
python3 - "$out" "$tag" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
f = sorted(glob.glob(f"{out}/bench/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
line = [ln for ln in open(f"{out}/bench.log") if ln.startswith("{")][-1]
d = json.loads(line)
L = ["# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline (whole default run: clip leg, bf16x3 leg, online legs, config 4)", "",
     "| kernel | calls | total ms | average ms | % of GPU time |", "|---|---|---|---|---|"]
for r in rows[:28]:
    n = r["Name"].replace("(anonymous namespace)::", "")
    n = n[: n.find("(")] if "(" in n else n
    L.append(f"| `{n.replace('void ', '')[:70]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e6:.4f} | {float(r['Percentage']):.2f} |")
ro = d["roofline"]
L += ["", f"Bench line of the same run: value {d['value']} {d['unit']}, ms_per_step {d['ms_per_step']}; roofline (HIP events around the "
      f"`tcn_stage_kernel` launches of the timed clip forwards): avg_launch_ms {ro.get('avg_launch_ms')}, achieved {ro['achieved']} TFLOP/s, frac {ro['frac']}.",
      "The `tcn_stage_kernel` rows above average over EVERY launch of the run (warm-up, A-GCN and bf16x3-leg launches of other shapes included);",
      f"the per-layer tables `{sys.argv[2]}_clip_layers.md` hold the timed-shape launches alone."]
open(f"{out}/bench_stats.md", "w").write("\n".join(L) + "\n")
print("\n".join(L[:14]))
PY
