#!/bin/bash
# device-only assembly listing of the product source with extra -D flags: tools/isa_probe/devasm.sh out.s [-D...]; then kstats.py / fstats.py out.s
out=$1; shift
cd "$(dirname "$0")/../../pbrt-v3-distributed_amd" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-spill-vgpr-to-agpr=0 -I../include -Icsrc --cuda-device-only -S "$@" csrc/pbrt_amd.hip -o $out 2>&1 | grep -v "hip-link"
