#!/usr/bin/env bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 --kernel-trace --stats of BASELINE configs[3] (Kinetics-400 shape, V = 18):
# (1) A-GCN clip forwards, batch 64 (tools/agcn_prof.py), (2) CoAGCN online, 1024 streams, one and two stream shards
# (tools/online_pass.py --model coagcn), and -- with "pmc" as second argument -- separate PMC passes (MFMA busy, FETCH /
# WRITE bytes) of the clip forward.  Summarise afterwards with
#   python tools/summarize_layers.py gpurun_out/prof_<tag>/agcn_clip  <tag>_agcn_clip_layers --mode clip --model agcn --batch 64 --cycles 4
#   python tools/summarize_layers.py gpurun_out/prof_<tag>/coagcn1 <tag>_coagcn_online_1shard --model agcn --shards 1
#   python tools/summarize_layers.py gpurun_out/prof_<tag>/coagcn2 <tag>_coagcn_online_2shards --model agcn --shards 2
# usage: bash tools/profile_config4.sh <tag> [pmc]
set -uo pipefail
tag="${1:-r03}"
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$R/gpurun_out/prof_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/agcn_clip" -- python3 "$R/tools/agcn_prof.py" 64 6 > "$out/agcn_clip.log" 2>&1
grep AGCN_PASS "$out/agcn_clip.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/coagcn1" -- python3 "$R/tools/online_pass.py" --model coagcn --shards 1 --cycles 16 > "$out/coagcn1.log" 2>&1
grep ONLINE_PASS "$out/coagcn1.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/coagcn2" -- python3 "$R/tools/online_pass.py" --model coagcn --shards 2 --cycles 16 > "$out/coagcn2.log" 2>&1
grep ONLINE_PASS "$out/coagcn2.log"
if [ "${2:-}" = pmc ]; then
  SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 GRBM_GUI_ACTIVE"
  rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d "$out/agcn_sq" -- python3 "$R/tools/agcn_prof.py" 64 3 > "$out/agcn_sq.log" 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/agcn_fetch" -- python3 "$R/tools/agcn_prof.py" 64 3 > "$out/agcn_fetch.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/agcn_write" -- python3 "$R/tools/agcn_prof.py" 64 3 > "$out/agcn_write.log" 2>&1
fi
find "$out" -name "*agent_info.csv" -delete
du -sh "$out"
