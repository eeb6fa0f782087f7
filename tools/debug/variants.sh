#!/bin/bash
# A/B runs of kernel variants built into pbrt-v3-distributed_amd/lib/variants/*.so
cd /root/repo
for v in "$@"; do
  PBRT_AMD_DEVICE_LIB=/root/repo/pbrt-v3-distributed_amd/lib/variants/$v.so timeout 300 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$v', d['value'], d['kernel_ms_per_step'])"
done
