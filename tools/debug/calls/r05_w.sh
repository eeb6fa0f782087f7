#!/bin/bash
# round 5, GPU call w: after call v -- the volumetric shading kernels back on twelve blocks per CU (one round had cost the subsurface C3 6 % of its shading), the hot-node test with
# a 2 % bound on the any-hit fetch counts; subsurface and smoke box at full size with their crops, the whole GPU suite once more.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q -n 4 > $O/r05_w_pytest.txt 2>&1; tail -3 $O/r05_w_pytest.txt
F="--steps 2 --warmup 1 --traffic none --cpu-port-seconds 0 --cpu-seconds 10"
timeout 500 python bench.py --subsurface $F 2> $O/r05_w_sss.err | tail -1 > $O/r05_w_bench_c3_subsurface.json
timeout 500 python bench.py --smokebox $F 2> $O/r05_w_smoke.err | tail -1 > $O/r05_w_bench_c3_smokebox.json
python - <<'EOF2'
import json
for c in ("c3_subsurface", "c3_smokebox"):
    try:
        d=json.load(open('/root/repo/gpurun_out/r05_w_bench_%s.json' % c)); print(c, d['value'], d['ms_per_step'], d['kernel_ms_per_step'], (d.get('cpu_baseline') or {}).get('parity_crop', {}).get('pixels_within_tol'))
    except Exception as e: print(c, 'ERR', e)
EOF2
