#!/bin/bash
# round 5, GPU call ab (the round's last): PT_PICK_EARLY=1 as the shipped default (call aa: k_shade 131.1 -> 128.6 ms per C3 frame) -- the whole GPU suite, smoke and the default bench line
# on the default library.  Anything but "202 passed" here = the source goes back to the state of call z.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 240 python -m pytest tests -m gpu -q -n 4 > $O/r05_ab_pytest.txt 2>&1; echo "pytest rc $?"; tail -2 $O/r05_ab_pytest.txt
timeout 100 python __graft_entry__.py smoke 2>&1 | tail -1 | tee -a $O/r05_ab_pytest.txt
timeout 400 python bench.py --save-traffic 2> $O/r05_ab_c3.err | tail -1 > $O/r05_ab_bench_c3.json; echo "default bench rc $?"
cp profiles/traffic_closest.json $O/r05_ab_traffic_closest.json
python - <<'EOF2'
import json
d=json.load(open('/root/repo/gpurun_out/r05_ab_bench_c3.json')); r=d['roofline']
print('c3', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], 'frac', r['frac'], (d['cpu_baseline'] or {}).get('parity_crop', {}).get('pixels_within_tol'), 'secondary', d['secondary']['textured_leafmask']['value'], d['secondary']['textured_leafmask']['parity_crop']['pixels_within_tol'])
EOF2
