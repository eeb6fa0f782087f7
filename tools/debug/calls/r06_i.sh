#!/bin/bash
# round 6 call i: (1) GPU suite serial -x with the fourth traversal mode (reference tree + hot nodes); (2) the default line with roofline_shade; (3) phase profile of
# k_shade<TEX> with Material::Bump separated (probe 21) on the textured + masked C3, 16 spp
cd /root/repo; O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/r06_i_pytest.txt 2>&1; tail -3 $O/r06_i_pytest.txt
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/r06_i_bench_c3.json 2> $O/r06_i_bench_c3.err
python -c "
import json; d=json.loads(open('$O/r06_i_bench_c3.json').read().strip().splitlines()[-1]); print('c3:', d['value'], d['kernel_ms_per_step'], 'frac', d['roofline']['frac']); print('roofline_shade:', d.get('roofline_shade'))"
PBRT_AMD_DEVICE_LIB=/root/repo/pbrt-v3-distributed_amd/lib/variants/shadeprof.so timeout 600 python bench.py --textured --leafmask --spp 16 --steps 1 --warmup 1 --traffic none --cpu-seconds 0 --secondary off > $O/r06_i_prof_tex.json 2> $O/r06_i_prof_tex.err
python -c "
import json; d=json.loads(open('$O/r06_i_prof_tex.json').read().strip().splitlines()[-1]); print('tex under the profiler:', d['kernel_ms_per_step'])"
grep "shade-prof" $O/r06_i_prof_tex.err | tail -24
