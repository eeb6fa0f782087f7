#!/bin/bash
# round 2: (1) GPU suite, (2) the extra-loads experiment (are L1 look-ups the bound of the node step?), (3) the new bench line end to end
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee gpurun_out/r02c_pytest.txt
run() { timeout 200 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none 2>gpurun_out/r02c_$1.err | tee gpurun_out/r02c_$1.json | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('$1', d['value'], d['kernel_ms_per_step'], 'nodes/ray', round(r['nodes_per_ray'],2), 'tris/ray', round(r['tris_per_ray'],2))"; }
run fast | tee gpurun_out/r02c_ab.txt
PBRT_AMD_DEVICE_LIB=/root/repo/pbrt-v3-distributed_amd/lib/variants/extraloads.so run extraloads | tee -a gpurun_out/r02c_ab.txt
timeout 900 python bench.py > gpurun_out/r02c_bench.json 2> gpurun_out/r02c_bench.err; tail -c 3000 gpurun_out/r02c_bench.json; tail -5 gpurun_out/r02c_bench.err
