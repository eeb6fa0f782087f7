#!/bin/bash
# round 6 call b: (1) the new GPU tests (gather cache, mixed shards, RCCL probe, one-rank RCCL, 2-rank image, cut-off scene); (2) VERDICT r5 item 7: what bit-identity
# of the libm calls and IEEE division costs -- the shipped library against two MEASUREMENT builds (lib/variants/fastlibm.so: hardware sin / cos / exp / log;
# fastall.so: + approximate fp32 division and sqrt, whole library), each with its own pbrt_ref crop check, on C3 / C2 / C4.
cd /root/repo; mkdir -p gpurun_out
# the multi-rank tests FIRST, on the cold box (the condition of the driver's round-5 run): how long does the torch warm-up take, do the jobs finish?
( time python -c "import torch" ) > gpurun_out/r06_b_cold_import.txt 2>&1
( time python -c "import torch; t = torch.zeros(1 << 20, device='cuda'); t += 1; torch.cuda.synchronize()" ) >> gpurun_out/r06_b_cold_import.txt 2>&1
timeout 2400 python -m pytest tests/test_zz_multirank_gpu.py -x -q -s -m gpu --durations=10 > gpurun_out/r06_b_pytest_multirank_cold.txt 2>&1
tail -3 gpurun_out/r06_b_pytest_multirank_cold.txt
timeout 900 python -m pytest tests -x -q -m gpu -k "film_gather" > gpurun_out/r06_b_pytest_new.txt 2>&1
tail -3 gpurun_out/r06_b_pytest_new.txt
V=/root/repo/pbrt-v3-distributed_amd/lib/variants
for cfg in c3 c2 c4; do
  for lib in shipped fastlibm fastall; do
    if [ $lib = shipped ]; then unset PBRT_AMD_DEVICE_LIB; else export PBRT_AMD_DEVICE_LIB=$V/$lib.so; fi
    timeout 900 python bench.py --config $cfg --steps 3 --warmup 1 --traffic none --secondary off --cpu-seconds 8 --cpu-port-seconds 0 > gpurun_out/r06_b_fm_${cfg}_$lib.json 2> gpurun_out/r06_b_fm_${cfg}_$lib.err
    echo "$cfg $lib rc $?: $(python - <<P
import json
try:
    d=json.loads(open('gpurun_out/r06_b_fm_${cfg}_$lib.json').read().strip().splitlines()[-1]); pc=(d.get('cpu_baseline') or {}).get('parity_crop') or {}
    print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms', {k: v for k, v in d['kernel_ms_per_step'].items()}, 'crop', pc.get('pixels_within_tol'), pc.get('relMSE'), pc.get('pixels'))
except Exception as e:
    print('no line', e)
P
)" | tee -a gpurun_out/r06_b_fast_math_study.txt
  done
done
unset PBRT_AMD_DEVICE_LIB
