#!/usr/bin/env bash
set -uo pipefail
tag="${1:-r06e}"
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$R/gpurun_out/prof_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/coagcn1" -- python3 "$R/tools/online_pass.py" --model coagcn --shards 1 --cycles 16 > "$out/coagcn1.log" 2>&1
grep ONLINE_PASS "$out/coagcn1.log"
cd "$R"
python3 tools/summarize_layers.py "$out/coagcn1" "${tag}_coagcn_online_1shard" --model agcn --shards 1 --cycles 16 > "gpurun_out/${tag}_coagcn_online_1shard.md" 2> "gpurun_out/${tag}_summarize.err" || tail -5 "gpurun_out/${tag}_summarize.err"
find "$out" -name "*agent_info.csv" -delete; find "$out" -name "*kernel_trace.csv" -delete
sed -n 1,70p "gpurun_out/${tag}_coagcn_online_1shard.md" | cut -c1-220
