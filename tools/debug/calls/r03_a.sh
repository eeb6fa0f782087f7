#!/bin/bash
# round 3, GPU call a: the parity suite on the glibc-identical device libm, then every BASELINE config at its quoted size (row g):
# C3 (default line + rocprofv3 kernel stats of the same command), C4 and C2 with their pbrt_ref crops and live roofline, C5 (4K / 512 spp on
# ONE GPU: 32 passes of 2^27 paths) with a pbrt_ref crop at all 512 spp, and the per-rank cost of the C5 frame for N = 2 / 4 / 8.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/r03_a_pytest.txt 2>&1; tail -3 $O/r03_a_pytest.txt
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2 | tee -a $O/r03_a_pytest.txt
timeout 600 python bench.py --save-traffic 2> $O/r03_a_c3.err | tail -1 > $O/r03_a_bench_c3.json; head -c 600 $O/r03_a_bench_c3.json; echo
cp profiles/traffic_closest.json $O/r03_a_traffic_closest.json
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/r03_a_prof -o c3 --output-format csv -- python $R/bench.py --cpu-seconds 0 --traffic none > $O/r03_a_bench_c3_under_rocprof.json 2> $O/r03_a_prof.err)
timeout 600 python bench.py --config c4 --steps 2 2> $O/r03_a_c4.err | tail -1 > $O/r03_a_bench_c4.json; head -c 300 $O/r03_a_bench_c4.json; echo
timeout 400 python bench.py --config c2 2> $O/r03_a_c2.err | tail -1 > $O/r03_a_bench_c2.json; head -c 300 $O/r03_a_bench_c2.json; echo
timeout 900 python bench.py --config c5 --steps 1 --warmup 1 --cpu-seconds 30 2> $O/r03_a_c5.err | tail -1 > $O/r03_a_bench_c5.json; head -c 300 $O/r03_a_bench_c5.json; echo
timeout 400 python tools/debug/scale_model.py --res 3840 2160 --spp 512 --reps 1 > $O/r03_a_scale_model_c5.json 2> $O/r03_a_scale.err; cat $O/r03_a_scale_model_c5.json
python - <<'EOF'
import json,glob
for f in sorted(glob.glob('/root/repo/gpurun_out/r03_a_bench_*.json')):
    try:
        d=json.load(open(f)); r=d.get('roofline') or {}; c=d.get('cpu_baseline') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'frac', r.get('frac'), 'launch ms', r.get('avg_launch_ms'), 'crop', (c.get('parity_crop') or {}), 'cpu', c.get('value'))
        print('   kernels', d.get('kernel_ms_per_step'))
    except Exception as e: print(f, 'ERR', e)
EOF
ls $O/r03_a_prof 2>/dev/null | head
