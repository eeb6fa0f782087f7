#!/bin/bash
# round 3, GPU call m: phase profile of k_shade on the final kernels (variant build -DPT_SHADE_PROF=1: wave time between consecutive probes, all
# memory drained at each probe) on the 16-spp C3 frame and on a 32-spp C4 frame.  Phase labels: see the PROBE(k) comments in k_shade.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/pbrt-v3-distributed_amd/lib/variants/shadeprof.so
PBRT_AMD_DEVICE_LIB=$V timeout 300 python bench.py --spp 16 --steps 1 --warmup 1 --cpu-seconds 0 --traffic none > $O/r03_m_c3.json 2> $O/r03_m_c3.err; grep "shade-prof" $O/r03_m_c3.err | tail -14 | tee $O/r03_m_shade_phase_profile_c3.txt
PBRT_AMD_DEVICE_LIB=$V timeout 300 python bench.py --config c4 --spp 32 --steps 1 --warmup 1 --cpu-seconds 0 --traffic none > $O/r03_m_c4.json 2> $O/r03_m_c4.err; grep "shade-prof" $O/r03_m_c4.err | tail -14 | tee $O/r03_m_shade_phase_profile_c4.txt
