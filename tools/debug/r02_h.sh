#!/bin/bash
# round 2: GPU suite with the final defaults; device fuzz of instanced scenes (two-level default); occupancy of the default traversal; the distinct-extra-loads experiment
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp; R=/root/repo
timeout 500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/r02h_pytest.txt
timeout 300 python tools/fuzz_vs_reference.py --device --instanced-only --n 700 --seed 11 --keep gpurun_out/fuzz_inst 2>&1 | tail -8 | tee gpurun_out/r02h_fuzz_inst.txt
timeout 200 python tools/fuzz_vs_reference.py --device --n 80 --seed 5 --keep gpurun_out/fuzz_dev 2>&1 | tail -8 | tee gpurun_out/r02h_fuzz.txt
run() { timeout 200 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none 2>gpurun_out/r02h_$1.err | tee gpurun_out/r02h_$1.json | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('$1', d['value'], d['kernel_ms_per_step'])"; }
run general_grid6 | tee gpurun_out/r02h_ab.txt
PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/grid5.so run general_grid5 | tee -a gpurun_out/r02h_ab.txt
PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/grid4.so run general_grid4 | tee -a gpurun_out/r02h_ab.txt
PBRT_AMD_TRACE=bvh4 run bvh4 | tee -a gpurun_out/r02h_ab.txt
PBRT_AMD_TRACE=bvh4 PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/extraloads3.so run bvh4_plus3loads | tee -a gpurun_out/r02h_ab.txt
