#!/usr/bin/env bash
set -u
CSK_DIAG=1 CSK_GCN16=2 timeout 900 python -m pytest tests/test_gpu_continual_parity.py tests/test_gpu_clip_parity.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_continual_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4
python tools/ab_env_sweep.py "" "CSK_STACK16=1" 2>&1 | grep AB_SWEEP
bash tools/r06_prof_online.sh r06d 1 > gpurun_out/r06_run5_prof.log 2>&1
sed -n '/Per layer/,/whole-config/p' gpurun_out/r06d_online_1shard.md
