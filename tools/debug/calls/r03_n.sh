#!/bin/bash
# round 3, GPU call n: the BSDF's lobe header (PT_LOBE_HEADER: lobe count + types of a material in one scalar load instead of a chain of dependent
# scalar loads through the 100-byte lobe records) -- parity suite, A/B on the 16-spp C3 probe frame, a 32-spp C4 frame and the 16-spp textured frame.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/r03_n_pytest.txt 2>&1; tail -3 $O/r03_n_pytest.txt
run() { tag=$1; cfg=$2; shift; shift; env "$@" timeout 300 python bench.py $cfg --steps 3 --warmup 1 --cpu-seconds 0 --traffic none 2>$O/r03_n_$tag.err | python -c "
import json,sys
d=json.load(sys.stdin)
print('$tag', d['value'], d['kernel_ms_per_step'])" | tee -a $O/r03_n_ab.txt; }
V=$R/pbrt-v3-distributed_amd/lib/variants/nolobehdr.so
run c3_cur "--spp 16" A=1
run c3_nolobehdr "--spp 16" PBRT_AMD_DEVICE_LIB=$V
run c3_cur_again "--spp 16" A=1
run c4_cur "--config c4 --spp 32" A=1
run c4_nolobehdr "--config c4 --spp 32" PBRT_AMD_DEVICE_LIB=$V
run c2_cur "--config c2 --spp 32" A=1
run c2_nolobehdr "--config c2 --spp 32" PBRT_AMD_DEVICE_LIB=$V
