#!/bin/bash
# Per-kernel resources of the device library as hipcc compiles it for gfx950 (no GPU needed): LDS bytes, VGPRs, scratch bytes per lane, kernel name -- one line each.
#   tools/debug/kernel_resources.sh out.txt [extra hipcc flags]
# Used in round 6 to find the kernels that reach a noise call (their group segment grows by the 256-byte permutation table) and to watch register / scratch budgets.
OUT=$(realpath -m $1); shift
cd "$(dirname "$0")/../../pbrt-v3-distributed_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-spill-vgpr-to-agpr=0 -I../include -Icsrc "$@" -S --cuda-device-only csrc/pbrt_amd.hip -o /tmp/pbrt_amd_$$.s 2>/dev/null
python3 - /tmp/pbrt_amd_$$.s > $OUT <<'P'
import re,sys,subprocess
txt=open(sys.argv[1]).read()
md=txt[txt.index('amdhsa.kernels:'):]
for blk in md.split('  - .agpr_count:')[1:]:
    name=re.search(r'\.name:\s+(\S+)',blk).group(1)
    g=re.search(r'\.group_segment_fixed_size:\s+(\d+)',blk).group(1)
    v=re.search(r'\.vgpr_count:\s+(\d+)',blk).group(1)
    sp=re.search(r'\.private_segment_fixed_size:\s+(\d+)',blk).group(1)
    dn=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip()
    print(g,v,sp,dn.split('(')[0])
P

