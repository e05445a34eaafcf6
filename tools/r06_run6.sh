#!/usr/bin/env bash
set -u
for sh in 1 2 3; do
python tools/ab_env_sweep.py --model coagcn --shards $sh --rounds 5 "" "CSK_STEP16=1" 2>&1 | grep AB_SWEEP
done
python tools/ab_env_sweep.py --shards 2 --rounds 5 "" "CSK_STEP16=1" 2>&1 | grep AB_SWEEP
