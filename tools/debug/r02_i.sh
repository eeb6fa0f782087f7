#!/bin/bash
# round 2: segmented queue counters (same-address atomics) + quantised BVH4 nodes (4 requests per step): GPU suite, then the 16 spp probe A/B
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp; R=/root/repo
timeout 500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/r02i_pytest.txt
run() { timeout 200 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none 2>gpurun_out/r02i_$1.err | tee gpurun_out/r02i_$1.json | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('$1', d['value'], d['kernel_ms_per_step'], 'nodes/ray', round(r['nodes_per_ray'],2), 'tris/ray', round(r['tris_per_ray'],2))"; tail -1 gpurun_out/r02i_$1.err | cut -c1-200; }
run general | tee gpurun_out/r02i_ab.txt
PBRT_AMD_TRACE=bvh4q run bvh4q | tee -a gpurun_out/r02i_ab.txt
PBRT_AMD_TRACE=bvh4 run bvh4lean | tee -a gpurun_out/r02i_ab.txt
