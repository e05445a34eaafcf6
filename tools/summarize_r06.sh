#!/usr/bin/env bash
# Condense the traces of tools/profile_r06.sh (gpurun_out/prof_<tag>/, merged back from the GPU box) into profiles/<tag>_*.md/.csv.
# Runs anywhere (pure Python over the CSVs).  usage: bash tools/summarize_r05.sh [tag]
set -euo pipefail
tag="${1:-r06}"
R="$(cd "$(dirname "$0")/.." && pwd)"
d="$R/gpurun_out/prof_$tag"
cd "$R"
python3 tools/summarize_layers.py "$d/clip_f32" "${tag}_clip_layers" --mode clip --cycles 3 > /dev/null
python3 tools/summarize_layers.py "$d/clip_bf16x3" "${tag}_clip_bf16x3_layers" --mode clip --cycles 3 > /dev/null
python3 tools/summarize_layers.py "$d/online1" "${tag}_online_1shard" --shards 1 > /dev/null
python3 tools/summarize_layers.py "$d/online2" "${tag}_online_2shards" --shards 2 > /dev/null
python3 tools/summarize_layers.py "$d/agcn_clip" "${tag}_agcn_clip_layers" --mode clip --model agcn --batch 64 --cycles 4 > /dev/null
for sh in 1 2 3; do
  python3 tools/summarize_layers.py "$d/coagcn$sh" "${tag}_coagcn_online_${sh}shard$([ $sh -gt 1 ] && echo s || true)" --model agcn --shards $sh --cycles 12 > /dev/null
done
python3 tools/summarize_pmc.py "$tag" "$tag" > /dev/null
cp "$d/bench_stats.md" "profiles/${tag}_rocprof_summary.md"
{ echo "# LDS counters of the online launches (${tag}): rocprofv3 --kernel-trace --pmc SQ_LDS_* -- tools/online_pass.py --shards 1 (CoST-GCN, 1024 streams)"; echo; cat "$d/lds_online_costgcn.md"; } > "profiles/${tag}_lds_online.md"
for st in 1 16; do
  { echo "# Few-stream latency path, rocprofv3 --kernel-trace --stats (${tag}_latency_${st}stream): tools/latency_pass.py --streams $st"; echo
    grep "ms per frame" "$d/latency$st.log" | sed 's/$/ (under the profiler)/'; echo; echo '```'; python3 tools/kstats.py "$d/latency$st" | head -24; echo '```'; } \
    > "profiles/${tag}_latency_${st}stream$([ $st -gt 1 ] && echo s || true).md"
done
for b in 1 8; do
  { echo "# Small-batch clip latency, rocprofv3 --kernel-trace --stats (${tag}_clip_latency_b$b): tools/clip_latency_pass.py --batch $b --split-k 4 --trace"; echo
    echo "StGcn clip forward, NTU-60 shape, batch $b, set_latency_mode(4): 3 warm-up + 20 eager forwards; every launch of the run (23 forwards)."; echo
    echo '```'; python3 tools/kstats.py "$d/clip_latency_b$b" | head -24; echo '```'; } > "profiles/${tag}_clip_latency_b$b.md"
done
ls -la profiles/${tag}_*
