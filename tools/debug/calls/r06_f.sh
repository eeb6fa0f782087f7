#!/bin/bash
# round 6 call f: sub-probes inside BSDF::Sample_f / f (profiler build), plain C3 at 16 spp; the texture tests + the textured line on the library with the fbm change
cd /root/repo; O=gpurun_out; mkdir -p $O
PBRT_AMD_DEVICE_LIB=/root/repo/pbrt-v3-distributed_amd/lib/variants/shadeprof.so timeout 600 python bench.py --spp 16 --steps 1 --warmup 1 --traffic none --cpu-seconds 0 --secondary off > $O/r06_f_prof_plain.json 2> $O/r06_f_prof_plain.err
python -c "
import json; d=json.loads(open('$O/r06_f_prof_plain.json').read().strip().splitlines()[-1]); print('plain under the profiler:', d['kernel_ms_per_step'])"
grep "shade-prof" $O/r06_f_prof_plain.err | tail -24
timeout 900 python -m pytest tests -x -q -m gpu -k "texture or textured or tex_ or alpha or edge_cases" > $O/r06_f_pytest_tex.txt 2>&1; tail -2 $O/r06_f_pytest_tex.txt
timeout 900 python bench.py --textured --leafmask --steps 3 --warmup 1 --traffic none --cpu-seconds 8 --cpu-port-seconds 0 --secondary off > $O/r06_f_bench_c3_textured_leafmask.json 2> $O/r06_f_bench_c3_textured_leafmask.err
python -c "
import json; d=json.loads(open('$O/r06_f_bench_c3_textured_leafmask.json').read().strip().splitlines()[-1]); print('textured+leafmask:', d['value'], d['kernel_ms_per_step'], d['cpu_baseline']['parity_crop'])"
