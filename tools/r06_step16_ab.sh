#!/usr/bin/env bash
# round 6: first measurement of the slot-balanced 16x16x4 TCN step tiles (step16.hip) against the 32x32x2 kernels
set -u
out=gpurun_out/r06_step16_ab.log
: > $out
./tools/microbench/bin/mfma16_order_probe >> $out 2>&1
echo "--- parity, new kernels forced on every K = 9 unsplit launch" >> $out
CSK_DIAG=1 CSK_STEP16=2 timeout 900 python -m pytest tests/test_gpu_continual_parity.py -x -q 2>&1 | tail -5 >> $out
for sh in 1 2; do
  echo "--- shards $sh" >> $out
  CSK_DIAG=1 CSK_STEP16=1 python tools/online_pass.py --shards $sh 2>&1 | grep ONLINE_PASS >> $out
  CSK_DIAG=1 CSK_STEP16=1 python tools/online_pass.py --shards $sh --no-fuse 2>&1 | grep ONLINE_PASS >> $out
  CSK_DIAG=1 CSK_STEP16=1 python tools/online_pass.py --shards $sh --no-fuse --force-ksplit 1 2>&1 | grep ONLINE_PASS >> $out
  echo "new TCN tiles (policy), unfused:" >> $out
  python tools/online_pass.py --shards $sh --no-fuse 2>&1 | grep ONLINE_PASS >> $out
  python tools/online_pass.py --shards $sh --no-fuse --force-ksplit 1 2>&1 | grep ONLINE_PASS >> $out
done
cat $out
