#!/bin/bash
# round 2: PT_TEX_SHADE_WAVES=2 (textured + volumetric shading kernels at 2 waves per SIMD / 256 VGPRs) vs the default 3
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp; R=/root/repo
run() { timeout 200 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none $2 2>gpurun_out/r02s_$1.err | tee gpurun_out/r02s_$1.json | python -c "
import json,sys
d=json.load(sys.stdin); print('$1', d['value'], d['kernel_ms_per_step'])"; }
run volpath_w3 --volpath | tee gpurun_out/r02s_ab.txt
PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/tex2.so run volpath_w2 --volpath | tee -a gpurun_out/r02s_ab.txt
PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/tex2.so run textured_w2 --textured | tee -a gpurun_out/r02s_ab.txt
PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/tex2.so run plain_with_tex2_lib | tee -a gpurun_out/r02s_ab.txt
PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/tex2.so timeout 300 python -m pytest tests -m gpu -x -q -k "volpath or textured or edge_cases or instancing" 2>&1 | tail -2 | tee -a gpurun_out/r02s_ab.txt
