#!/bin/bash
# round 2: phase profile of the TEXTURED shading kernel (k_shade<..., TEX>) on the textured C3 frame
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp; R=/root/repo
PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/shadeprof.so timeout 300 python bench.py --textured --spp 16 --steps 1 --warmup 1 --cpu-seconds 0 --cpu-port-seconds 0 --traffic none > gpurun_out/r02p_tex_prof.json 2> gpurun_out/r02p_tex_prof.err
grep shade-prof gpurun_out/r02p_tex_prof.err | tail -14
