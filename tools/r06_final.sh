#!/usr/bin/env bash
# round 6, final evidence run: full GPU suite (parity report), profile passes, bench stats
set -u
rm -f gpurun_out/parity_report.jsonl
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r06_final_tests.log
cat gpurun_out/r06_final_tests.log
bash tools/profile_r06.sh r06 > gpurun_out/profile_r06.log 2>&1
bash tools/profile_bench_stats.sh r06 >> gpurun_out/profile_r06.log 2>&1
grep -E "CLIP_PASS|ONLINE_PASS|AGCN_PASS|ms per frame" gpurun_out/profile_r06.log
python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err
tail -c 300 gpurun_out/r06_bench_line.json
