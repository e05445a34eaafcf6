#!/usr/bin/env bash
# Run ON THE GPU BOX: same-box A/B of library builds (build/variants/lib_<name>.so, e.g. different -DCSK_READ_AHEAD=n): each variant is
# copied over continual-skeletons_amd/libcskel_hip.so and timed in processes of its own -- the online path (tools/online_pass.py:
# CoST-GCN, 1024 streams) and the clip forward (tools/clip_pass.py: ST-GCN, batch 256) -- two interleaved rounds.
# usage: bash tools/ab_lib_variants.sh [online|clip|both]
set -u
what="${1:-both}"
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
cp continual-skeletons_amd/libcskel_hip.so /tmp/lib_orig.so
for round in 1 2; do
  for v in build/variants/lib_*.so; do
    cp "$v" continual-skeletons_amd/libcskel_hip.so
    o=""; c=""
    [ "$what" != clip ] && o=$(python tools/online_pass.py --cycles 48 2>&1 | grep ONLINE_PASS | sed -E "s/.*(ms_per_cycle=[0-9.]+ frames_per_s=[0-9]+).*/\1/")
    [ "$what" != online ] && c=$(python tools/clip_pass.py --forwards 6 2>&1 | grep CLIP_PASS | cut -c1-160)
    echo "VARIANT $(basename "$v") round $round: $o | $c"
  done
done
cp /tmp/lib_orig.so continual-skeletons_amd/libcskel_hip.so
