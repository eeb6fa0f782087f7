#!/bin/bash
# round 2: row f4 second device run (wavefront form of homogeneous volpath scenes, two-level instancing in k_shade_vol), trig A/B, k_shade phase profile
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp; R=/root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02l_pytest.txt; tail -3 gpurun_out/r02l_pytest.txt
(timeout 300 python tools/fuzz_vs_reference.py --device --media --n 80 --seed 61 2>&1 | tail -3; timeout 300 python tools/fuzz_vs_reference.py --device --media --sss --n 80 --seed 62 2>&1 | tail -3) > gpurun_out/r02l_fuzz.txt 2>&1; cat gpurun_out/r02l_fuzz.txt
timeout 600 python bench.py --volpath --spp 16 --steps 2 --warmup 1 --cpu-seconds 10 --cpu-port-seconds 0 --traffic none > gpurun_out/r02l_bench_volpath.json 2> gpurun_out/r02l_bench_volpath.err; python -c "
import json; d=json.load(open('gpurun_out/r02l_bench_volpath.json')); print('volpath 16spp', d['value'], d['kernel_ms_per_step'], d['cpu_baseline']['value'], d['cpu_baseline']['parity_crop'])"
run() { timeout 200 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none 2>gpurun_out/r02l_$1.err | tee gpurun_out/r02l_$1.json | python -c "
import json,sys
d=json.load(sys.stdin); print('$1', d['value'], d['kernel_ms_per_step'])"; }
run default | tee gpurun_out/r02l_ab.txt
PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/ocmltrig.so run ocmltrig | tee -a gpurun_out/r02l_ab.txt
PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/shadeprof.so run shadeprof | tee -a gpurun_out/r02l_ab.txt
grep shade-prof gpurun_out/r02l_shadeprof.err | tail -14
