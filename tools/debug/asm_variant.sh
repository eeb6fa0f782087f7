#!/bin/bash
# Developer aid: an A/B build of the device library with a post-pass over the COMPILER'S ASSEMBLY (never built by build(); the product is the plain
# hipcc build).  usage: tools/debug/asm_variant.sh NAME 'sed-script' [extra -D flags]   ->  pbrt-v3-distributed_amd/lib/variants/NAME.so
# Steps = what `hipcc -###` runs for csrc/pbrt_amd.hip, with the device half split at the assembly: device -S, sed, assemble, lld, bundle, host half
# with the patched fat binary embedded.  Used for ISA-level questions tools/valu_probe/issue_probe raised (which encodings cost what in the real kernels).
set -e
NAME=$1; SED=$2; shift 2 || true
R=/root/repo/pbrt-v3-distributed_amd; W=/tmp/asm_variant_$NAME; L=/opt/rocm/lib/llvm/bin
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-spill-vgpr-to-agpr=0 -I$R/../include -I$R/csrc $*"
mkdir -p $W $R/lib/variants
cd $R
if [ ! -f $W/dev.s ] || [ csrc/pbrt_amd.hip -nt $W/dev.s ] || [ -n "$*" ]; then /opt/rocm/bin/hipcc $FLAGS --cuda-device-only -S csrc/pbrt_amd.hip -o $W/dev.s 2>/dev/null; fi
sed -E "$SED" $W/dev.s > $W/patched.s
echo "[asm_variant] $NAME: $(diff $W/dev.s $W/patched.s | grep -c '^>') lines changed"
$L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $W/patched.s -o $W/dev.o
$L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $W/dev.out $W/dev.o
$L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/dev.out -output=$W/dev.hipfb
/opt/rocm/bin/hipcc $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/dev.hipfb -c csrc/pbrt_amd.hip -o $W/host.o 2>/dev/null
/opt/rocm/bin/hipcc -shared $W/host.o -o lib/variants/$NAME.so
ls -la lib/variants/$NAME.so
