#!/bin/bash
# round 4, GPU call l: device-vs-oracle fuzz campaign on the round's final kernels (TravNodeStepQ2 / TravStackB in every quantised-node instance: closest / any hit / MIS,
# spheres, alpha masks, walked segments; the flattened TriangleTest everywhere; k_sss_probe_tail) -- the quantised single-level kernels run the scenes whose instances are
# flattened (PBRT_AMD_INSTANCING=0), the two-level and volumetric kernels the others.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
F=$O/r04_l_device_fuzz_campaign.txt; : > $F
fz() { echo "== $*  ${ARGS}" | tee -a $F; env "$@" timeout 600 python tools/fuzz_vs_reference.py --device ${ARGS} 2>&1 | tail -2 | tee -a $F; }
ARGS="--n 300 --seed 701" fz PBRT_AMD_INSTANCING=0
ARGS="--n 120 --seed 702" fz A=1
ARGS="--media --sss --n 150 --seed 703" fz A=1
ARGS="--media --n 100 --seed 704" fz PBRT_AMD_INSTANCING=0
