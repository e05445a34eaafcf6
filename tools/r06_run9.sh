#!/usr/bin/env bash
set -u
CSK_DIAG=1 CSK_GCN16=2 timeout 900 python -m pytest tests/test_gpu_continual_parity.py tests/test_gpu_clip_parity.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -3
python tools/stamp16_probe.py 2>&1 | grep "STAMP16 gcn" | cut -c1-330
python tools/gcn16_skip_probe.py 2>&1 | grep GCN16_SKIP
python tools/online_pass.py --shards 1 2>&1 | grep ONLINE_PASS
python tools/online_pass.py --shards 1 2>&1 | grep ONLINE_PASS
