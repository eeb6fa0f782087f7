#!/bin/bash
# round 3, GPU call q: the state the round ends with (after the lobe header and the walked interfaces; call k was the same set one step earlier) -- parity suite (agreement figures written out) + smoke, every BASELINE config at its quoted
# size with its pbrt_ref crop and live roofline (C3 = the default line, with the live VALU-issue figure; C2, C4, C5), rocprofv3 kernel stats of
# the default command, per-rank cost of the C3 frame for N = 2 / 4 / 8.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
rm -f $O/r03_q_parity_report.jsonl
PBRT_AMD_PARITY_REPORT=$O/r03_q_parity_report.jsonl timeout 900 python -m pytest tests -m gpu -x -q > $O/r03_q_pytest.txt 2>&1; tail -3 $O/r03_q_pytest.txt
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2 | tee -a $O/r03_q_pytest.txt
timeout 700 python bench.py --save-traffic 2> $O/r03_q_c3.err | tail -1 > $O/r03_q_bench_c3.json
cp profiles/traffic_closest.json $O/r03_q_traffic_closest.json
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/r03_q_prof -o c3 --output-format csv -- python $R/bench.py --cpu-seconds 0 --traffic none > $O/r03_q_bench_c3_under_rocprof.json 2> $O/r03_q_prof.err)
head -5 $O/r03_q_prof/c3_kernel_stats.csv | cut -c1-150
timeout 400 python bench.py --config c2 --cpu-seconds 10 --cpu-port-seconds 0 2> $O/r03_q_c2.err | tail -1 > $O/r03_q_bench_c2.json
timeout 600 python bench.py --config c4 --steps 2 --cpu-seconds 10 --cpu-port-seconds 0 2> $O/r03_q_c4.err | tail -1 > $O/r03_q_bench_c4.json
timeout 900 python bench.py --config c5 --steps 1 --warmup 1 --cpu-seconds 30 --cpu-port-seconds 0 2> $O/r03_q_c5.err | tail -1 > $O/r03_q_bench_c5.json
timeout 300 python tools/debug/scale_model.py > $O/r03_q_scale_model_c3.json 2> $O/r03_q_scale.err; cat $O/r03_q_scale_model_c3.json
python - <<'EOF2'
import json
for c in ("c3", "c2", "c4", "c5"):
    try:
        d=json.load(open('/root/repo/gpurun_out/r03_q_bench_%s.json' % c)); r=d['roofline']
        print(c, d['value'], d['ms_per_step'], d['kernel_ms_per_step'], 'launch ms', round(r['avg_launch_ms'], 2), 'frac', r['frac'], 'valu', (r.get('valu_issue') or {}).get('issue_slots_frac'), (d.get('cpu_baseline') or {}).get('parity_crop', {}).get('pixels_within_tol'), (d.get('cpu_baseline') or {}).get('value'))
    except Exception as e: print(c, 'ERR', e)
EOF2
timeout 500 python bench.py --fogbox --cpu-seconds 10 --cpu-port-seconds 0 --traffic none 2> $O/r03_q_fogbox.err | tail -1 > $O/r03_q_bench_c3_fogbox.json
timeout 400 python bench.py --volpath --cpu-seconds 0 --traffic none 2> $O/r03_q_haze.err | tail -1 > $O/r03_q_bench_c3_volpath.json
python - <<'EOF2'
import json
for c in ("c3_fogbox", "c3_volpath"):
    try:
        d=json.load(open('/root/repo/gpurun_out/r03_q_bench_%s.json' % c))
        print(c, d['value'], d['ms_per_step'], d['kernel_ms_per_step'], (d.get('cpu_baseline') or {}).get('parity_crop', {}).get('pixels_within_tol'))
    except Exception as e: print(c, 'ERR', e)
EOF2
