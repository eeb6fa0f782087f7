#!/bin/bash
# round 6 call e: the spatial light pick's first fetches issued early (round 5's patch, held back then for a stall that was the 2-rank test's cold start).
# (1) GPU suite serial -x; (2) the default line; (3) the contention-free phase profile of k_shade (plain C3 and textured + masked C3, 16 spp)
cd /root/repo; O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/r06_e_pytest.txt 2>&1; tail -3 $O/r06_e_pytest.txt
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/r06_e_bench_c3.json 2> $O/r06_e_bench_c3.err; head -c 300 $O/r06_e_bench_c3.json; echo
for w in plain tex; do
  if [ $w = tex ]; then X="--textured --leafmask"; else X=""; fi
  PBRT_AMD_DEVICE_LIB=/root/repo/pbrt-v3-distributed_amd/lib/variants/shadeprof.so timeout 600 python bench.py $X --spp 16 --steps 1 --warmup 1 --traffic none --cpu-seconds 0 --secondary off > $O/r06_e_prof_$w.json 2> $O/r06_e_prof_$w.err
  python -c "
import json; d=json.loads(open('$O/r06_e_prof_$w.json').read().strip().splitlines()[-1]); print('$w under the profiler:', d['kernel_ms_per_step'])"
  grep "shade-prof" $O/r06_e_prof_$w.err | tail -16
done
