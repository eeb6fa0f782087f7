#!/bin/bash
# round 2: BVH4Q default -- variants (entry distances on the stack, 4 blocks per CU, 256-ray fetch batches), then the full default bench line
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp; R=/root/repo
run() { timeout 200 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none 2>gpurun_out/r02j_$1.err | tee gpurun_out/r02j_$1.json | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('$1', d['value'], d['kernel_ms_per_step'], 'nodes/ray', round(r['nodes_per_ray'],2), 'tris/ray', round(r['tris_per_ray'],2))"; }
run bvh4q | tee gpurun_out/r02j_ab.txt
for v in stackt grid4 batch256; do PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/$v.so run $v | tee -a gpurun_out/r02j_ab.txt; done
timeout 900 python bench.py --save-traffic > gpurun_out/r02j_bench.json 2> gpurun_out/r02j_bench.err; tail -c 3200 gpurun_out/r02j_bench.json; tail -3 gpurun_out/r02j_bench.err | cut -c1-300
cp profiles/traffic_closest.json gpurun_out/r02j_traffic_closest.json
