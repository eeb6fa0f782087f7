#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
run() { timeout 300 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 2>gpurun_out/s4_$1.err | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('$1', d['value'], d['kernel_ms_per_step'], 'nodes/ray', round(r['nodes_per_ray'],2), 'tris/ray', round(r['tris_per_ray'],2))"; }
run cur
for v in "$@"; do PBRT_AMD_DEVICE_LIB=/root/repo/pbrt-v3-distributed_amd/lib/variants/$v.so run $v; done
