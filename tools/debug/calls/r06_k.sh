#!/bin/bash
# round 6 call k: a longer device-vs-oracle fuzz campaign on the round's kernels (seeds not used before): plain, two-level, media + subsurface, tile-serial samplers, spectra
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
F=$O/r06_k_device_fuzz_campaign.txt; : > $F
fz() { echo "== $*  ${ARGS}" | tee -a $F; env "$@" timeout 900 python tools/fuzz_vs_reference.py --device ${ARGS} 2>&1 | tail -2 | tee -a $F; }
ARGS="--n 400 --seed 911" fz PBRT_AMD_INSTANCING=0
ARGS="--n 300 --seed 912" fz A=1
ARGS="--media --sss --n 300 --seed 913" fz A=1
ARGS="--media --n 200 --seed 914" fz PBRT_AMD_INSTANCING=0
ARGS="--pixel-samplers --n 120 --seed 915" fz A=1
ARGS="--spectra --n 150 --seed 916" fz A=1
ARGS="--instanced-only --n 200 --seed 917" fz A=1
