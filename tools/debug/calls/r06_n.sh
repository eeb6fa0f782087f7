#!/bin/bash
# round 6 call n: a third device-vs-oracle fuzz campaign on the round's kernels, seeds not used before (GPU minutes that would otherwise lapse)
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
F=$O/r06_n_device_fuzz_campaign.txt; : > $F
fz() { echo "== $*  ${ARGS}" | tee -a $F; env "$@" timeout 1500 python tools/fuzz_vs_reference.py --device ${ARGS} 2>&1 | tail -2 | tee -a $F; }
ARGS="--n 700 --seed 921" fz PBRT_AMD_INSTANCING=0
ARGS="--n 600 --seed 922" fz A=1
ARGS="--media --sss --n 500 --seed 923" fz A=1
ARGS="--media --n 400 --seed 924" fz PBRT_AMD_INSTANCING=0
ARGS="--pixel-samplers --n 200 --seed 925" fz A=1
ARGS="--spectra --n 300 --seed 926" fz A=1
ARGS="--instanced-only --n 400 --seed 927" fz A=1
