#!/usr/bin/env bash
# Build libcskel_hip.so (the C ABI of include/cskel.h) for gfx950.  Cross-compiles without a GPU.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../libcskel_hip.so"
hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -I"$here/../../include" \
      "$here"/*.hip -o "$out" "$@"
echo "built $out"
