#!/usr/bin/env bash
# Resource check of every gfx950 kernel of the library after a kernel change (no GPU needed): kernels that use scratch memory
# (a closure or an array the compiler could not keep in registers -- DESIGN.md section 3.1c lists how that happened silently)
# and, for the conv kernels, the register count that decides the resident waves per SIMD.   bash tools/check_kernels.sh [file.hip ...]
set -u
R="$(cd "$(dirname "$0")/.." && pwd)"; cd "$R/yolact_minimal_amd/csrc"
FILES="${*:-$(ls *.hip)}"
bad=0
for f in $FILES; do
    out=$(/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I"$R/include" -I. -fhip-fp32-correctly-rounded-divide-sqrt \
          --cuda-device-only -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1)
    echo "$out" | awk -v file="$f" '
        /Function Name:/ { name = $0; sub(/.*Function Name: /, "", name); sub(/ \[-Rpass.*/, "", name) }
        /VGPRs:/ && !/Spill/ { v = $0; sub(/.*VGPRs: /, "", v); sub(/ .*/, "", v) }
        /AGPRs:/ { a = $0; sub(/.*AGPRs: /, "", a); sub(/ .*/, "", a) }
        /ScratchSize/ { s = $0; sub(/.*: /, "", s); sub(/ .*/, "", s);
                        if (s + 0 > 0) { printf("SCRATCH %5d B/lane  %s  %s\n", s, file, name); } n++ }
        END { printf("%-18s %3d kernels checked\n", file, n) }'
    if echo "$out" | grep -q "ScratchSize \[bytes/lane\]: [1-9]"; then bad=1; fi
done
exit $bad
