#!/usr/bin/env bash
# kernel trace of the online path (1 stream shard) + per-layer summary -> gpurun_out/<tag>_online_1shard.md
set -uo pipefail
tag="${1:-r06a}"; shards="${2:-1}"; shift 2 || true
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$R/gpurun_out/prof_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/online$shards" -- python3 "$R/tools/online_pass.py" --shards $shards "$@" > "$out/online$shards.log" 2>&1
grep ONLINE_PASS "$out/online$shards.log"
cd "$R"
python3 tools/summarize_layers.py "$out/online$shards" "${tag}_online_${shards}shard" --shards $shards > "gpurun_out/${tag}_online_${shards}shard.md" 2> "gpurun_out/${tag}_summarize.err" || tail -5 "gpurun_out/${tag}_summarize.err"
python3 tools/kstats.py "$out/online$shards" 14
find "$out" -name "*agent_info.csv" -delete; find "$out" -name "*kernel_trace.csv" -delete
sed -n 1,60p "gpurun_out/${tag}_online_${shards}shard.md"
