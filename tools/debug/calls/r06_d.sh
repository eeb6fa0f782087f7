#!/bin/bash
# round 6 call d: where does k_shade<TEX> spend its time on the textured + masked C3 (16 spp)?  per-phase wave cycles (PT_SHADE_PROF build)
cd /root/repo; O=gpurun_out; mkdir -p $O
PBRT_AMD_DEVICE_LIB=/root/repo/pbrt-v3-distributed_amd/lib/variants/shadeprof.so timeout 600 python bench.py --textured --leafmask --spp 16 --steps 1 --warmup 1 --traffic none --cpu-seconds 0 --secondary off > $O/r06_d_prof_tex.json 2> $O/r06_d_prof_tex.err
grep "shade-prof" $O/r06_d_prof_tex.err | tail -24
PBRT_AMD_DEVICE_LIB=/root/repo/pbrt-v3-distributed_amd/lib/variants/shadeprof.so timeout 600 python bench.py --spp 16 --steps 1 --warmup 1 --traffic none --cpu-seconds 0 --secondary off > $O/r06_d_prof_plain.json 2> $O/r06_d_prof_plain.err
grep "shade-prof" $O/r06_d_prof_plain.err | tail -24
