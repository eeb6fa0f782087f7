#!/bin/bash
# round 2: grain of the dynamic item distribution in k_shade / k_shade_vol; request-rate ladder
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp; R=/root/repo
run() { timeout 200 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none $2 2>gpurun_out/r02o_$1.err | tee gpurun_out/r02o_$1.json | python -c "
import json,sys
d=json.load(sys.stdin); print('$1', d['value'], d['kernel_ms_per_step'])"; }
run default | tee gpurun_out/r02o_ab.txt
python -c "
import json; d=json.load(open('gpurun_out/r02o_default.json')); print(d['roofline'].get('request_rate'))" | tee -a gpurun_out/r02o_ab.txt
for v in dyn64 dyn128 dyn256 dyn512; do PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/$v.so run $v | tee -a gpurun_out/r02o_ab.txt; done
run volpath_static --volpath | tee -a gpurun_out/r02o_ab.txt
PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/dyn256.so run volpath_dyn256 --volpath | tee -a gpurun_out/r02o_ab.txt
PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/dyn256.so timeout 300 python -m pytest tests -m gpu -x -q -k "volpath or textured or edge_cases" 2>&1 | tail -3 | tee gpurun_out/r02o_pytest.txt
