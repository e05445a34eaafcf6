#!/usr/bin/env bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 --kernel-trace --stats of (1) the online shape of bench.py, single stream
# shard, (2) the same with two stream shards, (3) the clip workload.  Summarise afterwards (here or in the container) with
#   python tools/summarize_layers.py gpurun_out/prof_<tag>/online1 <tag>_online_1shard --shards 1
#   python tools/summarize_layers.py gpurun_out/prof_<tag>/online2 <tag>_online_2shards --shards 2
#   python tools/summarize_layers.py gpurun_out/prof_<tag>/clip    <tag>_clip_layers --mode clip --cycles 3
# usage: bash tools/profile_layers.sh <tag> [what: all|online|clip]
set -uo pipefail
tag="${1:-r02}"
what="${2:-all}"
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$R/gpurun_out/prof_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
if [ "$what" != clip ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/online1" -- python3 "$R/tools/online_pass.py" --shards 1 > "$out/online1.log" 2>&1
  grep ONLINE_PASS "$out/online1.log"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/online2" -- python3 "$R/tools/online_pass.py" --shards 2 > "$out/online2.log" 2>&1
  grep ONLINE_PASS "$out/online2.log"
fi
if [ "$what" != online ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/clip" -- python3 "$R/bench.py" --workload clip --steps 4 --warmup 1 --no-cpu-baseline > "$out/clip.log" 2>&1
  tail -1 "$out/clip.log" | cut -c1-300
fi
# keep what travels back small: per-dispatch traces + stats only
find "$out" -name "*agent_info.csv" -delete
du -sh "$out"
