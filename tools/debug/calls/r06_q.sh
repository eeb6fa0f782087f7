#!/bin/bash
# round 6 call q: the closest-hit / any-hit instances of SPHERE scenes (killeroo-simple.pbrt as shipped = C2) in the MID shape (two 512-thread blocks per CU, hot nodes in LDS) --
# GPU suite, then C2: shipped (MID), PBRT_AMD_HOT=0 (MID shape without hot nodes), lib/variants/midmis.so (the MIS instance in the MID shape as well)
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R; T=r06_q
timeout 1500 python -m pytest tests -x -q -m gpu --durations=5 > $O/${T}_pytest.txt 2>&1; tail -2 $O/${T}_pytest.txt
line() { python - "$1" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); pc=(d.get('cpu_baseline') or {}).get('parity_crop') or {}
    print(sys.argv[1].split('/')[-1], d['value'], 'Msamples/s', d['ms_per_step'], 'ms', d['kernel_ms_per_step'], 'crop', pc.get('pixels_within_tol'), pc.get('pixels'))
except Exception as e: print(sys.argv[1], 'no line', e)
P
}
B="--config c2 --steps 3 --warmup 1 --traffic none --secondary off --cpu-port-seconds 0"
timeout 900 python bench.py $B --cpu-seconds 8 > $O/${T}_bench_c2.json 2> $O/${T}_bench_c2.err; line $O/${T}_bench_c2.json
PBRT_AMD_HOT=0 timeout 900 python bench.py $B --cpu-seconds 0 > $O/${T}_bench_c2_hot0.json 2> $O/${T}_bench_c2_hot0.err; line $O/${T}_bench_c2_hot0.json
PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/midmis.so timeout 900 python bench.py $B --cpu-seconds 8 > $O/${T}_bench_c2_midmis.json 2> $O/${T}_bench_c2_midmis.err; line $O/${T}_bench_c2_midmis.json
timeout 900 python bench.py $B --cpu-seconds 0 > $O/${T}_bench_c2_again.json 2> $O/${T}_bench_c2_again.err; line $O/${T}_bench_c2_again.json
