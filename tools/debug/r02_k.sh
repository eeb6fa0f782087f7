#!/bin/bash
# round 2: first device run of row f4 (k_shade_vol: volpath + BSSRDF) -- the whole -m gpu suite, device fuzzing of volumetric / subsurface
# scenes against the oracle, the C3 stand-in in a homogeneous medium under "volpath", and the default frame as a regression check
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02k_pytest.txt; tail -3 gpurun_out/r02k_pytest.txt
export PBRT_AMD_INSTANCING=0
(timeout 400 python tools/fuzz_vs_reference.py --device --media --n 80 --seed 41 2>&1 | tail -4; timeout 400 python tools/fuzz_vs_reference.py --device --media --sss --n 80 --seed 42 2>&1 | tail -4) > gpurun_out/r02k_fuzz.txt; cat gpurun_out/r02k_fuzz.txt
unset PBRT_AMD_INSTANCING
timeout 600 python bench.py --volpath --spp 16 --steps 2 --warmup 1 --cpu-seconds 10 --cpu-port-seconds 0 --traffic none > gpurun_out/r02k_bench_volpath.json 2> gpurun_out/r02k_bench_volpath.err; tail -c 2500 gpurun_out/r02k_bench_volpath.json; tail -3 gpurun_out/r02k_bench_volpath.err | cut -c1-300
timeout 300 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none > gpurun_out/r02k_bench16.json 2> gpurun_out/r02k_bench16.err; python -c "
import json; d=json.load(open('gpurun_out/r02k_bench16.json')); print('default 16spp', d['value'], d['kernel_ms_per_step'])"
