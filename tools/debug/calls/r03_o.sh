#!/bin/bash
# round 3, GPU call o: wavefront form for scenes with BSDF-less interfaces between homogeneous media (DevVol::tr_queues, k_vol_tr) -- parity suite
# (vol_inst takes the new form two-level and flattened), device fuzz of random media scenes, the C3 stand-in with a bank of fog behind a BSDF-less
# box (bench.py --fogbox): the new form against the general form (PBRT_AMD_VOL_TR_QUEUES=0), with a pbrt_ref crop.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/r03_o_pytest.txt 2>&1; tail -3 $O/r03_o_pytest.txt
F=$O/r03_o_device_fuzz_media.txt; : > $F
fz() { echo "== $*" | tee -a $F; env "$@" timeout 600 python tools/fuzz_vs_reference.py --device ${ARGS} 2>&1 | tail -2 | tee -a $F; }
ARGS="--media --n 250 --seed 811" fz A=1
ARGS="--media --n 250 --seed 812" fz PBRT_AMD_INSTANCING=0
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --fogbox --spp 16 --steps 2 --warmup 1 --cpu-seconds ${CPUS:-0} --cpu-port-seconds 0 --traffic none 2>$O/r03_o_$tag.err | tail -1 > $O/r03_o_bench_fogbox_$tag.json; python -c "
import json
d=json.load(open('$O/r03_o_bench_fogbox_$tag.json'))
print('$tag', d['value'], d['kernel_ms_per_step'], (d.get('cpu_baseline') or {}).get('parity_crop'))" | tee -a $O/r03_o_ab_fogbox_16spp.txt; }
CPUS=10 run tr_queues A=1
run general PBRT_AMD_VOL_TR_QUEUES=0
