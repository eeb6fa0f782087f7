#!/bin/bash
# round 6 call p: k_film lists the spilling samples (second launch walks the list only) on top of call o:
# GPU suite (serial), then C3 / textured C3 / C2 / C4 lines (3 steps, no CPU legs) and the default line with its crop.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R; T=${1:-r06_p}
timeout 1500 python -m pytest tests -x -q -m gpu --durations=5 > $O/${T}_pytest.txt 2>&1; tail -2 $O/${T}_pytest.txt
line() { python - "$1" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); pc=(d.get('cpu_baseline') or {}).get('parity_crop') or {}
    print(sys.argv[1].split('/')[-1], d['value'], 'Msamples/s', d['ms_per_step'], 'ms', d['kernel_ms_per_step'], 'crop', pc.get('pixels_within_tol'), pc.get('pixels'))
except Exception as e: print(sys.argv[1], 'no line', e)
P
}
for spec in "c3:" "c3_textured_leafmask:--textured --leafmask" "c2:--config c2" "c4:--config c4"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout 900 python bench.py $args --steps 3 --warmup 1 --traffic none --secondary off --cpu-seconds 8 --cpu-port-seconds 0 > $O/${T}_bench_$name.json 2> $O/${T}_bench_$name.err
  line $O/${T}_bench_$name.json
done
