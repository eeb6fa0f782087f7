#!/bin/bash
# round 2: ray binning before traversal -- GPU suite, then the C3 frame (16 spp probe) for the three traversal layouts with binning, one without
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp; R=/root/repo
timeout 400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/r02f_pytest.txt
run() { timeout 200 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic ${2:-none} 2>gpurun_out/r02f_$1.err | tee gpurun_out/r02f_$1.json | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('$1', d['value'], d['kernel_ms_per_step'], 'nodes/ray', round(r['nodes_per_ray'],2), 'tris/ray', round(r['tris_per_ray'],2), 'traffic', r.get('traffic'), 'frac', r.get('frac'))"; tail -1 gpurun_out/r02f_$1.err; }
PBRT_AMD_TRACE=bvh4 run bvh4_bin live | tee gpurun_out/r02f_ab.txt
run bvh8c_bin live | tee -a gpurun_out/r02f_ab.txt
PBRT_AMD_TRACE=general run general_bin | tee -a gpurun_out/r02f_ab.txt
PBRT_AMD_RAYBIN=0 PBRT_AMD_TRACE=bvh4 run bvh4_nobin | tee -a gpurun_out/r02f_ab.txt
