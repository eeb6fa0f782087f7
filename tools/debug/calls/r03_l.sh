#!/bin/bash
# round 3, GPU call l (REFUTED, code reverted; the variant it names no longer builds): k_shade warms the caches for the wave's next item (PT_SHADE_PREFETCH) -- parity suite, A/B on the 16-spp C3 probe frame and
# on a 32-spp C4 frame against the same build without it.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/r03_l_pytest.txt 2>&1; tail -3 $O/r03_l_pytest.txt
run() { tag=$1; cfg=$2; shift; shift; env "$@" timeout 300 python bench.py $cfg --steps 3 --warmup 1 --cpu-seconds 0 --traffic none 2>$O/r03_l_$tag.err | python -c "
import json,sys
d=json.load(sys.stdin)
print('$tag', d['value'], d['kernel_ms_per_step'])" | tee -a $O/r03_l_ab.txt; }
V=$R/pbrt-v3-distributed_amd/lib/variants/noprefetch.so
run c3_cur "--spp 16" A=1
run c3_noprefetch "--spp 16" PBRT_AMD_DEVICE_LIB=$V
run c3_cur_again "--spp 16" A=1
run c4_cur "--config c4 --spp 32" A=1
run c4_noprefetch "--config c4 --spp 32" PBRT_AMD_DEVICE_LIB=$V
run c3tex_cur "--textured --spp 16" A=1
run c3tex_noprefetch "--textured --spp 16" PBRT_AMD_DEVICE_LIB=$V
