#!/usr/bin/env bash
# Build libcskel_hip.so (the C ABI of include/cskel.h) for gfx950.  Cross-compiles without a GPU.
# The compiler's per-kernel resource report (registers, scratch = spills, LDS, occupancy) is kept next to the library
# as kernel_resources.txt; tests/test_host_logic_cpu.py::test_no_kernel_spills reads it.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../libcskel_hip.so"
hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -I"$here/../../include" \
      -Rpass-analysis=kernel-resource-usage "$here"/*.hip -o "$out" "$@" 2> "$here/../kernel_resources.raw" || {
    cat "$here/../kernel_resources.raw" >&2; exit 1; }
grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size" "$here/../kernel_resources.raw" \
    | sed -E 's/^.*remark: +//; s/^Function Name: /Name: /; s/ *\[-Rpass-analysis=kernel-resource-usage\]//' > "$here/../kernel_resources.txt" || true
grep -v "Rpass-analysis=kernel-resource-usage\|^ *[0-9]* *|\|^ *|\|\^\|remarks* generated" "$here/../kernel_resources.raw" >&2 || true
rm -f "$here/../kernel_resources.raw"
echo "built $out"
