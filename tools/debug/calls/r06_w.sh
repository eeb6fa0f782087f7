#!/bin/bash
# round 6 call w (call v again, bounded: v asked rocprofv3 for six TCC counters in one pass and never came back -- 20 GPU-minutes, no output):
# memory-side read traffic per kernel of one C3 frame with and without the merged triangle record (lib/variants/notrirec.so), from bench.py's own four-counter pass;
# then the 2-rank one-device job 30 x on HEAD and a device-vs-oracle fuzz campaign on new seeds.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
F=$O/r06_w_kernel_traffic.txt; : > $F
KT_TIMEOUT_S=300 timeout 330 python tools/debug/kernel_traffic.py "shipped (tri_rec)" 2>&1 | tail -14 | tee -a $F
PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/notrirec.so KT_TIMEOUT_S=300 timeout 330 python tools/debug/kernel_traffic.py "three per-triangle arrays (notrirec)" 2>&1 | tail -14 | tee -a $F
KT_TIMEOUT_S=300 timeout 330 python tools/debug/kernel_traffic.py "shipped, textured + masked" --textured --leafmask 2>&1 | tail -14 | tee -a $F
LIMIT=200 tools/debug/n2_loop.sh /root/repo 30 r06_w > $O/r06_w_n2_loop.txt 2>&1; tail -2 $O/r06_w_n2_loop.txt
F=$O/r06_w_device_fuzz_campaign.txt; : > $F
fz() { echo "== $*  ${ARGS}" | tee -a $F; env "$@" timeout 500 python tools/fuzz_vs_reference.py --device ${ARGS} 2>&1 | tail -2 | tee -a $F; }
ARGS="--n 220 --seed 911" fz PBRT_AMD_INSTANCING=0
ARGS="--n 90 --seed 912" fz A=1
ARGS="--media --sss --n 110 --seed 913" fz A=1
ARGS="--media --n 80 --seed 914" fz PBRT_AMD_INSTANCING=0
