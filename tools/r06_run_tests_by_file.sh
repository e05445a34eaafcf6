#!/usr/bin/env bash
# GPU suite file by file (a crash in the native library takes the whole pytest process down: this names the file and the test)
set -u
for f in tests/test_gpu_*.py; do
  timeout 1500 python -X faulthandler -m pytest "$f" -x -q -v 2>&1 > gpurun_out/t_$(basename $f .py).log
  rc=$?
  echo "== $f rc=$rc: $(grep -E "passed|failed|error" gpurun_out/t_$(basename $f .py).log | tail -1)"
  if [ $rc -ne 0 ]; then grep -E "PASSED|FAILED|ERROR|Fatal|fault|File \"/root" gpurun_out/t_$(basename $f .py).log | tail -12; fi
done
