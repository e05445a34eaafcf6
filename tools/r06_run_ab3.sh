#!/usr/bin/env bash
# round 6, late: (a) workgroup start / end distribution of the step kernels, (b) A-GCN clip with the 16-wide temporal conv forced
set -u
python tools/stamp16_probe.py > gpurun_out/stamp16.log 2>&1; grep STAMP16 gpurun_out/stamp16.log
python tools/ab_agcn_probe.py CSK_TCN16=2 2>&1 | tail -1 | tee gpurun_out/ab_agcn_tcn16.log
cd /tmp && export TMPDIR=/tmp
CSK_DIAG=1 CSK_TCN16=2 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_agcn_tcn16" -- python3 "$GRAFT_REPO_ROOT/tools/agcn_prof.py" 64 6 > "$GRAFT_REPO_ROOT/gpurun_out/agcn_tcn16.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python3 tools/summarize_layers.py gpurun_out/prof_agcn_tcn16 r06x_agcn_tcn16_layers --mode clip --model agcn --batch 64 --cycles 4 > /dev/null 2>gpurun_out/sum_err.log
ls profiles | grep r06x; grep "tcn_stage" profiles/r06x_agcn_tcn16_layers.md | head -20
cp profiles/r06x_agcn_tcn16_layers.md gpurun_out/ 2>/dev/null
