#!/bin/bash
# round 2: lean traversal steps (pt_trace_fast.h) -- full GPU suite, then A/B on the C3 frame (16 spp probe) and the default bench line
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/r02b_pytest.txt
run() { timeout 200 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 2>gpurun_out/r02b_$1.err | tee gpurun_out/r02b_$1.json | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('$1', d['value'], d['kernel_ms_per_step'], 'nodes/ray', round(r['nodes_per_ray'],2), 'tris/ray', round(r['tris_per_ray'],2))"; }
PBRT_AMD_TRACE=general run general | tee gpurun_out/r02b_ab.txt
run fast | tee -a gpurun_out/r02b_ab.txt
timeout 300 python bench.py --cpu-seconds 0 > gpurun_out/r02b_bench64.json 2> gpurun_out/r02b_bench64.err; tail -c 1800 gpurun_out/r02b_bench64.json
