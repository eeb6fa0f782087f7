#!/bin/bash
# round 4, GPU call t: the last 3 GPU minutes -- the library as committed at the end of the round (call q's kernels + the routing fix for mix materials; the EWA experiment reverted):
# smoke, the fixture / routing / texture parity tests, the default line once more without the counter passes.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 100 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $O/r04_t_smoke_and_tests.txt
timeout 100 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "only_the_vertices or fixture or tex" 2>&1 | tail -2 | tee -a $O/r04_t_smoke_and_tests.txt
timeout 100 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --cpu-port-seconds 0 --traffic none 2> $O/r04_t_c3.err | tail -1 > $O/r04_t_bench_c3_no_counter_passes.json
python -c "
import json; d=json.load(open('$O/r04_t_bench_c3_no_counter_passes.json')); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'])" | tee -a $O/r04_t_smoke_and_tests.txt
