#!/bin/bash
# GPU test suite, then the TEXTURED C3 stand-in (16 spp frame) with the built library and each lib/variants/<v>.so
cd /root/repo; mkdir -p gpurun_out
timeout 200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/abtex_pytest.txt
run() { timeout 120 python bench.py --textured --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 2>gpurun_out/abtex_$1.err | tee gpurun_out/abtex_$1.json | python -c "
import json,sys
d=json.load(sys.stdin)
print('$1', d['value'], d['kernel_ms_per_step'])"; }
run cur
for v in "$@"; do PBRT_AMD_DEVICE_LIB=/root/repo/pbrt-v3-distributed_amd/lib/variants/$v.so run $v; done
