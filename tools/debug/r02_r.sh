#!/bin/bash
# round 2: k_shade occupancy targets with the dynamic item distribution (2 / 3 / 4 / 5 waves per SIMD)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp; R=/root/repo
run() { timeout 200 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none $2 2>gpurun_out/r02r_$1.err | tee gpurun_out/r02r_$1.json | python -c "
import json,sys
d=json.load(sys.stdin); print('$1', d['value'], d['kernel_ms_per_step'])"; }
run sw3_default | tee gpurun_out/r02r_ab.txt
for v in sw2 sw4 sw5; do PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/$v.so run $v | tee -a gpurun_out/r02r_ab.txt; done
PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/sw4.so run sw4_textured --textured | tee -a gpurun_out/r02r_ab.txt
PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/sw2.so run sw2_textured --textured | tee -a gpurun_out/r02r_ab.txt
