#!/bin/bash
# round 2: direct-lighting traversals on a second stream (PBRT_AMD_OVERLAP=1), overlapping the next bounce's path-extension traversal
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { timeout 200 python bench.py --spp 16 --steps 3 --warmup 1 --cpu-seconds 0 --traffic none $2 2>gpurun_out/r02u_$1.err | tee gpurun_out/r02u_$1.json | python -c "
import json,sys
d=json.load(sys.stdin); print('$1', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"; }
run serial | tee gpurun_out/r02u_ab.txt
PBRT_AMD_OVERLAP=1 run overlap | tee -a gpurun_out/r02u_ab.txt
PBRT_AMD_OVERLAP=1 timeout 300 python -m pytest tests -m gpu -x -q -k "li_per_sample or render_image or baseline or edge_cases or textured or volpath or gather" 2>&1 | tail -2 | tee -a gpurun_out/r02u_ab.txt
