#!/bin/bash
# round 5, GPU call z: the committed state once more -- GPU suite, smoke, the default bench line (saved as the traffic file's source) -- and a device-vs-oracle fuzz campaign on
# the round's final kernels (interleaved shading partition, own topology, texture routines by reference / in-line nodes, mask records, MID traversal shape, hit-list tail).
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -n 4 > $O/r05_z_pytest.txt 2>&1; tail -2 $O/r05_z_pytest.txt
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -1 | tee -a $O/r05_z_pytest.txt
timeout 900 python bench.py --save-traffic 2> $O/r05_z_c3.err | tail -1 > $O/r05_z_bench_c3.json; echo "default bench rc $?"
cp profiles/traffic_closest.json $O/r05_z_traffic_closest.json
python - <<'EOF2'
import json
d=json.load(open('/root/repo/gpurun_out/r05_z_bench_c3.json')); r=d['roofline']
print('c3', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], 'frac', r['frac'], (d['cpu_baseline'] or {}).get('parity_crop', {}).get('pixels_within_tol'), 'secondary', d['secondary']['textured_leafmask']['value'], d['secondary']['textured_leafmask']['parity_crop']['pixels_within_tol'])
EOF2
F=$O/r05_z_device_fuzz_campaign.txt; : > $F
fz() { echo "== $*  ${ARGS}" | tee -a $F; env "$@" timeout 400 python tools/fuzz_vs_reference.py --device ${ARGS} 2>&1 | tail -2 | tee -a $F; }
ARGS="--n 220 --seed 801" fz PBRT_AMD_INSTANCING=0
ARGS="--n 90 --seed 802" fz A=1
ARGS="--media --sss --n 110 --seed 803" fz A=1
ARGS="--media --n 80 --seed 804" fz PBRT_AMD_INSTANCING=0
