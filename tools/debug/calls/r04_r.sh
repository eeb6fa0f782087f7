#!/bin/bash
# round 4, GPU call r (after call q; kernels unchanged, upload-side fix of the routing for mix materials built on subsurface materials): the subsurface / vol parity tests incl. the new
# sss_mix scene, the subsurface line again, and evidence for the next round -- phase profile of k_shade<..., TEX> on the textured + masked frame (variant build -DPT_SHADE_PROF=1; two
# new probes around the per-lane material evaluation), rocprofv3 kernel tables of the textured + masked and of the subsurface frame at 64 spp.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "subsurface or bssrdf or walked or sss or vol or only_the_vertices" 2>&1 | tail -2 | tee $O/r04_r_pytest_sss.txt
F="--steps 2 --warmup 1 --traffic none --cpu-port-seconds 0 --cpu-seconds 0"
timeout 300 python bench.py --subsurface $F 2> $O/r04_r_sss.err | tail -1 > $O/r04_r_bench_c3_subsurface.json
V=$R/pbrt-v3-distributed_amd/lib/variants/shadeprof.so
PBRT_AMD_DEVICE_LIB=$V timeout 300 python bench.py --textured --leafmask --spp 16 --steps 1 --warmup 1 --cpu-seconds 0 --cpu-port-seconds 0 --traffic none > $O/r04_r_texlm_prof.json 2> $O/r04_r_texlm_prof.err; grep "shade-prof" $O/r04_r_texlm_prof.err | tail -16 | tee $O/r04_r_shade_phase_profile_textured_leafmask.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/r04_r_prof_texlm -o texlm --output-format csv -- python $R/bench.py --textured --leafmask $F > $O/r04_r_bench_texlm_under_rocprof.json 2> $O/r04_r_prof_texlm.err)
head -12 $O/r04_r_prof_texlm/texlm_kernel_stats.csv | cut -c1-200
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/r04_r_prof_sss -o sss --output-format csv -- python $R/bench.py --subsurface $F > $O/r04_r_bench_sss_under_rocprof.json 2> $O/r04_r_prof_sss.err)
head -14 $O/r04_r_prof_sss/sss_kernel_stats.csv | cut -c1-200
python - <<'EOF2'
import json
for c in ("c3_subsurface",):
    d=json.load(open('/root/repo/gpurun_out/r04_r_bench_%s.json' % c)); print(c, d['value'], d['ms_per_step'], d['kernel_ms_per_step'])
EOF2
