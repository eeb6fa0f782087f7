#!/bin/bash
# round 6 call j: how many lanes does k_shade run with, phase by phase?  (profiler build, C3 at 16 spp)
cd /root/repo; O=gpurun_out; mkdir -p $O
PBRT_AMD_DEVICE_LIB=/root/repo/pbrt-v3-distributed_amd/lib/variants/shadeprof.so timeout 600 python bench.py --spp 16 --steps 1 --warmup 1 --traffic none --cpu-seconds 0 --secondary off > $O/r06_j_prof_plain.json 2> $O/r06_j_prof_plain.err
grep "shade-prof" $O/r06_j_prof_plain.err | tail -22
