#!/bin/bash
# round 2: GPU suite with textured spheres + batched sampler in k_shade_vol, volpath frame, scheduling sweep of the default traversal kernel
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp; R=/root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02m_pytest.txt; tail -3 gpurun_out/r02m_pytest.txt
timeout 600 python bench.py --volpath --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --cpu-port-seconds 0 --traffic none > gpurun_out/r02m_bench_volpath.json 2> gpurun_out/r02m_bench_volpath.err; python -c "
import json; d=json.load(open('gpurun_out/r02m_bench_volpath.json')); print('volpath 16spp', d['value'], d['kernel_ms_per_step'])"
run() { timeout 200 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none 2>gpurun_out/r02m_$1.err | tee gpurun_out/r02m_$1.json | python -c "
import json,sys
d=json.load(sys.stdin); print('$1', d['value'], d['kernel_ms_per_step'])"; }
run default | tee gpurun_out/r02m_ab.txt
for v in ns4 ns16 lm16 lm32 lm40 rf8 rf32 ns32lm32; do PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/$v.so run $v | tee -a gpurun_out/r02m_ab.txt; done
