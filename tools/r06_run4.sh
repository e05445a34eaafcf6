#!/usr/bin/env bash
set -u
timeout 1500 python -m pytest tests/test_gpu_continual_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -6
for i in 1 2; do
python tools/online_pass.py --shards 1 2>&1 | grep ONLINE_PASS
CSK_DIAG=1 CSK_STACK16=1 python tools/online_pass.py --shards 1 2>&1 | grep ONLINE_PASS
done
bash tools/r06_prof_online.sh r06c 1 > gpurun_out/r06_run4_prof.log 2>&1
sed -n '/Per kernel/,/whole-config/p' gpurun_out/r06c_online_1shard.md
bash tools/r06_pmc_online.sh r06p 2>&1 | tail -40
