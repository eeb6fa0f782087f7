#!/bin/bash
# round 3, GPU call j: (1) branch-free axis permutation in the watertight triangle test (PT_TRI_SELECT) -- parity suite (the reference's Triangle.*
# vectors bit for bit), A/B on the 16-spp C3 probe frame; (2) the per-lane tracer of k_shade_vol on the quantised nodes (PT_VOL_LANE_QN): the general
# form of the volpath frame (PBRT_AMD_VOL_INLINE=1), A/B against the 128-byte nodes.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/r03_j_pytest.txt 2>&1; tail -3 $O/r03_j_pytest.txt
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --spp 16 --steps 3 --warmup 1 --cpu-seconds 0 --traffic none 2>$O/r03_j_$tag.err | python -c "
import json,sys
d=json.load(sys.stdin)
print('$tag', d['value'], d['kernel_ms_per_step'])" | tee -a $O/r03_j_ab_16spp.txt; }
run cur A=1
run tribranch PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/tribranch.so
run cur_again A=1
vol() { tag=$1; shift; env "$@" timeout 400 python bench.py --volpath --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none 2>$O/r03_j_$tag.err | python -c "
import json,sys
d=json.load(sys.stdin)
print('$tag', d['value'], d['kernel_ms_per_step'])" | tee -a $O/r03_j_ab_volpath_16spp.txt; }
vol vol_wavefront A=1
vol vol_general_qn PBRT_AMD_VOL_INLINE=1
vol vol_general_128 PBRT_AMD_VOL_INLINE=1 PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/vollane128.so
