#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
run() { tag=$1; shift; timeout 300 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 "$@" 2>gpurun_out/s5_$tag.err | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('$tag', d['value'], d['kernel_ms_per_step'])"; }
run p64_64 --spp 64 --max-paths 67108864
run p134_64 --spp 64 --max-paths 134217728
rocm-smi --showmeminfo vram | head -8
