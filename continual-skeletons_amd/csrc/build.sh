#!/usr/bin/env bash
# Build libcskel_hip.so (the C ABI of include/cskel.h) for gfx950.  Cross-compiles without a GPU.
# Every .hip file is compiled to an object of its own (in parallel, rebuilt only when the file or a header changed) and the
# objects are linked into the library.  The compiler's per-kernel resource report (registers, scratch = spills, LDS,
# occupancy) is kept next to the library as kernel_resources.txt; tests/test_host_logic_cpu.py::test_no_kernel_spills reads it.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../libcskel_hip.so"
obj="$here/../../build/obj"
mkdir -p "$obj"
newest_hdr=$(ls -t "$here"/*.h "$here/../../include"/*.h | head -1)
compile_one() {
    src="$1"; o="$obj/$(basename "${src%.hip}").o"; raw="$o.raw"
    if [ -f "$o" ] && [ -f "$raw" ] && [ "$o" -nt "$src" ] && [ "$o" -nt "$newest_hdr" ] && [ "$o" -nt "$here/build.sh" ]; then exit 0; fi
    if ! hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I"$here/../../include" -Rpass-analysis=kernel-resource-usage \
           -c "$src" -o "$o" $EXTRA 2> "$raw"; then
        cat "$raw" >&2; rm -f "$o"; exit 1
    fi
}
export -f compile_one
export obj here newest_hdr EXTRA="$*"
ls "$here"/*.hip | xargs -P "$(nproc)" -I{} bash -c 'compile_one {}'
hipcc --offload-arch=gfx950 -shared -fPIC "$obj"/*.o -o "$out"
cat "$obj"/*.raw > "$here/../kernel_resources.raw"
grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size" "$here/../kernel_resources.raw" \
    | sed -E 's/^.*remark: +//; s/^Function Name: /Name: /; s/ *\[-Rpass-analysis=kernel-resource-usage\]//' > "$here/../kernel_resources.txt" || true
grep -v "Rpass-analysis=kernel-resource-usage\|^ *[0-9]* *|\|^ *|\|\^\|remarks* generated" "$here/../kernel_resources.raw" >&2 || true
rm -f "$here/../kernel_resources.raw"
echo "built $out"
