#!/bin/bash
# round 2: k_shade A/B -- the lobe's own f skipped in Sample_f (default) vs computed, dynamic per-wave item distribution at three grains, request-rate ceiling
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp; R=/root/repo
run() { timeout 200 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none 2>gpurun_out/r02n_$1.err | tee gpurun_out/r02n_$1.json | python -c "
import json,sys
d=json.load(sys.stdin); print('$1', d['value'], d['kernel_ms_per_step'], d['roofline'].get('request_rate'))"; }
run default | tee gpurun_out/r02n_ab.txt
for v in noskipf dyn1024 dyn256 dyn4096 dyn1024g3; do PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/$v.so run $v | tee -a gpurun_out/r02n_ab.txt; done
timeout 300 python -m pytest tests -m gpu -x -q -k "bxdf or li_per_sample or render_image or baseline" 2>&1 | tail -3 | tee gpurun_out/r02n_pytest.txt
PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/dyn1024.so timeout 300 python -m pytest tests -m gpu -x -q -k "li_per_sample or render_image or baseline or edge_cases" 2>&1 | tail -3 | tee -a gpurun_out/r02n_pytest.txt
