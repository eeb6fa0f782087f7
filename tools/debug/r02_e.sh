#!/bin/bash
# round 2: compressed 8-wide BVH (pt_bvh8c.h) -- GPU suite, A/B on the C3 frame (16 spp probe), FETCH_SIZE of the default
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp; R=/root/repo
timeout 400 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 | tee gpurun_out/r02e_pytest.txt
run() { timeout 200 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic ${2:-none} 2>gpurun_out/r02e_$1.err | tee gpurun_out/r02e_$1.json | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('$1', d['value'], d['kernel_ms_per_step'], 'nodes/ray', round(r['nodes_per_ray'],2), 'tris/ray', round(r['tris_per_ray'],2), 'traffic', r.get('traffic'), 'frac', r.get('frac'))"; tail -2 gpurun_out/r02e_$1.err; }
run bvh8c live | tee gpurun_out/r02e_ab.txt
PBRT_AMD_TRACE=bvh4 run bvh4 | tee -a gpurun_out/r02e_ab.txt
