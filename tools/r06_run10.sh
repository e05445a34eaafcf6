#!/usr/bin/env bash
set -u
timeout 1500 python -m pytest tests/test_gpu_clip_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_agcn_parity.py tests/test_gpu_precision_modes.py -x -q 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "clip or config5" 2>&1 | tail -3
for i in 1 2; do
CSK_DIAG=1 CSK_TCN16=1 python tools/clip_pass.py --precision f32 2>&1 | grep CLIP_PASS | sed 's/^/old: /'
python tools/clip_pass.py --precision f32 2>&1 | grep CLIP_PASS | sed 's/^/new: /'
done
CSK_DIAG=1 CSK_TCN16=1 python tools/agcn_prof.py 64 6 2>&1 | grep AGCN_PASS | sed 's/^/old: /'
python tools/agcn_prof.py 64 6 2>&1 | grep AGCN_PASS | sed 's/^/new: /'
