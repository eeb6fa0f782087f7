#!/bin/bash
# round 6 call r: k_trace<1> of sphere scenes in the MID shape by default; the sort's rank from one returning LDS atomic per lane (lib/variants/keyatomic.so) against the ballot rounds
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R; T=r06_r
timeout 1500 python -m pytest tests -x -q -m gpu --durations=5 > $O/${T}_pytest.txt 2>&1; tail -2 $O/${T}_pytest.txt
line() { python - "$1" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); pc=(d.get('cpu_baseline') or {}).get('parity_crop') or {}
    print(sys.argv[1].split('/')[-1], d['value'], 'Msamples/s', d['ms_per_step'], 'ms', d['kernel_ms_per_step'], 'crop', pc.get('pixels_within_tol'), pc.get('pixels'))
except Exception as e: print(sys.argv[1], 'no line', e)
P
}
B="--steps 3 --warmup 1 --traffic none --secondary off --cpu-port-seconds 0 --cpu-seconds 0"
for cfg in c3 c2 c4; do
  for lib in shipped keyatomic shipped2 keyatomic2; do
    case $lib in keyatomic*) export PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/keyatomic.so;; *) unset PBRT_AMD_DEVICE_LIB;; esac
    timeout 900 python bench.py --config $cfg $B > $O/${T}_bench_${cfg}_$lib.json 2> $O/${T}_bench_${cfg}_$lib.err; line $O/${T}_bench_${cfg}_$lib.json
  done
done
