#!/usr/bin/env bash
set -u
bash tools/r06_prof_online.sh r06b 1 > gpurun_out/r06_run3_prof.log 2>&1
head -3 gpurun_out/r06_run3_prof.log
sed -n '/Per layer/,/whole-config/p' gpurun_out/r06b_online_1shard.md
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15
