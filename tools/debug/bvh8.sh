#!/bin/bash
# First GPU check of the experimental BVH8 traversal (csrc/pt_bvh8.h, k_trace<..., WIDE>; compiled in round 1, never run on a GPU then):
# the image-level parity tests and a 16-spp C3 frame with PBRT_AMD_BVH8=1, next to the default BVH4 path.
cd /root/repo; mkdir -p gpurun_out
PBRT_AMD_BVH8=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "image or reference or fixture or radiance or sharded" 2>&1 | tail -5 | tee gpurun_out/bvh8_pytest.txt
run() { timeout 200 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 2>gpurun_out/bvh8_$1.err | tee gpurun_out/bvh8_$1.json | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('$1', d['value'], d['kernel_ms_per_step'], 'nodes/ray', round(r['nodes_per_ray'],2), 'tris/ray', round(r['tris_per_ray'],2))"; }
run bvh4
PBRT_AMD_BVH8=1 run bvh8
