#!/bin/bash
# DEVELOPER TOOL: compile pbrt-v3-distributed_amd/csrc for the x86 host against tools/hostemu/shim (see README.md).
# Output: tools/hostemu/_build/libpbrt_amd_hostemu.so (git-ignored, gpurun-ignored).  Never built by build(), never loaded by the package.
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd)
B=$HERE/_build; mkdir -p $B/src
cp $ROOT/pbrt-v3-distributed_amd/csrc/*.h $ROOT/pbrt-v3-distributed_amd/csrc/*.inc $B/src/
cp $ROOT/pbrt-v3-distributed_amd/csrc/pbrt_amd.hip $B/src/pbrt_amd.cpp
# register-liveness asm, one-thread blocks, one-lane waves, dynamic LDS, address-space qualifiers
sed -i -E 's/asm volatile\([^;]*\);//g; s/__attribute__\(\(address_space\([0-9]\)\)\)//g' $B/src/*.h $B/src/pbrt_amd.cpp
sed -i -E 's/^#define PT_BLOCK 256/#define PT_BLOCK 1/; ' $B/src/pt_scene.h
sed -i -E '/^PT_DEV const mi_bxdf \*Generic\(const mi_bxdf \*b\) \{ return b; \}/d' $B/src/pt_shade.h   # same signature as the constant-address-space overload once the qualifier is gone
sed -i -E 's/extern __shared__ uint32_t lhist\[\];/uint32_t *lhist = (uint32_t *)emu_dyn_lds;/' $B/src/pbrt_amd.cpp
CXX=${EMU_CXX:-/opt/rocm/lib/llvm/bin/clang++}
$CXX -x c++ -std=c++17 ${EMU_OPT:--O1} -g -fPIC -shared -ffp-contract=off -Wno-everything \
  -DPT_TRACEQ_BLOCK=1 -DPT_TRACE_MID_BLOCK=1 -DTRACE_REFILL=1 -DTRACE_LEAF_MIN=1 -DTRACE_BATCH=1u -DPT_HOST_EMU=1 -DPT_WAVE_SIZE=1u ${EMU_DEFS} -DPT_WAVE_APPEND3=emu_wave_append3 \
  -I$HERE/shim -I$ROOT/include -I$B/src $B/src/pbrt_amd.cpp $HERE/emu_globals.cpp -o $B/libpbrt_amd_hostemu.so -lpthread -ldl
echo built $B/libpbrt_amd_hostemu.so
