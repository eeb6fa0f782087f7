#!/bin/bash
# round 4, GPU call v (the round's last GPU seconds): wave-uniform texture / material evaluation in k_shade<..., TEX> (PT_TEX_UNIFORM=1, lib/variants/texu.so; the shipped
# library is the per-lane form, instruction for instruction the build of call t): texture / fixture parity on the variant, then the textured frame at 16 spp (per-lane form in call s: 165.8).
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
export PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/texu.so
timeout 25 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tex or fixture" 2>&1 | tail -1 | tee $O/r04_v_texu.txt
timeout 40 python bench.py --textured --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --cpu-port-seconds 0 --traffic none 2> $O/r04_v.err | tail -1 > $O/r04_v_bench_tex_uniform.json
python -c "
import json; d=json.load(open('$O/r04_v_bench_tex_uniform.json')); print('texu', d['value'], d['ms_per_step'], {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})" | tee -a $O/r04_v_texu.txt
