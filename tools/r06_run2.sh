#!/usr/bin/env bash
set -u
out=gpurun_out/r06_run2.log
: > $out
echo "--- full GPU suite, default policy" >> $out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 >> $out
echo "--- continual + fullsize + clip parity with gcn16 forced" >> $out
CSK_DIAG=1 CSK_GCN16=2 timeout 900 python -m pytest tests/test_gpu_continual_parity.py tests/test_gpu_clip_parity.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -5 >> $out
for sh in 1 2; do
  echo "--- shards $sh: old family / new" >> $out
  CSK_DIAG=1 CSK_STEP16=1 python tools/online_pass.py --shards $sh --force-ksplit 3 2>&1 | grep ONLINE_PASS >> $out
  python tools/online_pass.py --shards $sh 2>&1 | grep ONLINE_PASS >> $out
  CSK_DIAG=1 CSK_GCN16=1 python tools/online_pass.py --shards $sh 2>&1 | grep ONLINE_PASS >> $out
done
cat $out
