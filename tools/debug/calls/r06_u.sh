#!/bin/bash
# round 6 call u: LoadHitTriangle (one 128-byte line per hit triangle) also in the volumetric / subsurface shading kernels: GPU suite, subsurface + smoke box + fog lines with crops, C3 default line
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R; T=r06_u
timeout 1500 python -m pytest tests -x -q -m gpu --durations=5 > $O/${T}_pytest.txt 2>&1; tail -2 $O/${T}_pytest.txt
line() { python - "$1" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); pc=(d.get('cpu_baseline') or {}).get('parity_crop') or {}
    print(sys.argv[1].split('/')[-1], d['value'], 'Msamples/s', d['ms_per_step'], 'ms', d['kernel_ms_per_step'], 'crop', pc.get('pixels_within_tol'), pc.get('pixels'))
except Exception as e: print(sys.argv[1], 'no line', e)
P
}
B="--steps 3 --warmup 1 --traffic none --secondary off --cpu-port-seconds 0 --cpu-seconds 8"
for spec in "c3:" "sss:--subsurface" "smoke:--smokebox" "fog:--fogbox" "haze:--volpath"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout 900 python bench.py $args $B > $O/${T}_bench_${name}.json 2> $O/${T}_bench_${name}.err; line $O/${T}_bench_${name}.json
done
