#!/usr/bin/env bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for bench.py.
# usage: bash tools/profile.sh <tag>      -> writes gpurun_out/prof_<tag>/{trace,fetch,write,sq}/...
set -uo pipefail
tag="${1:-r01}"
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$R/gpurun_out/prof_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 3 --warmup 1 --step-cycles 8 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 $B > "$out/trace.log" 2>&1
tail -1 "$out/trace.log" | cut -c1-400
# HBM traffic: FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots), each with --kernel-trace only.
# PMC passes run the CLIP workload alone so that per-launch averages are not mixed with step-mode launches.
BC="$B --workload clip"
BS="$B --workload step"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 $BC > "$out/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python3 $BC > "$out/write.log" 2>&1
# same two passes on the online path: 4-frame launches over 1024 streams only (tools/step_traffic_pass.py)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch_step" -- python3 "$R/tools/step_traffic_pass.py" > "$out/fetch_step.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write_step" -- python3 "$R/tools/step_traffic_pass.py" > "$out/write_step.log" 2>&1
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d "$out/sq" -- python3 $BC > "$out/sq.log" 2>&1
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d "$out/sq_step" -- python3 $BS > "$out/sq_step.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/lds" -- python3 $BC > "$out/lds.log" 2>&1
find "$out" -name "*.csv" | wc -l
