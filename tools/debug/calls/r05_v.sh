#!/bin/bash
# round 5, GPU call v: the state the round ends with (as call q, after the mask records, the parallel key scan and the one-round shading grid) -- parity suite + smoke, the default bench line (live request-size counters, SQ pass, pbrt_ref crop, secondary.textured_leafmask),
# rocprofv3 kernel stats of the default workload, every other BASELINE config at its quoted size with its crop, and the variants of C3 (textured + leaf masks, subsurface, smoke box).
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
rm -f $O/r05_v_parity_report.jsonl
PBRT_AMD_PARITY_REPORT=$O/r05_v_parity_report.jsonl timeout 1200 python -m pytest tests -m gpu -q -n 4 > $O/r05_v_pytest.txt 2>&1; tail -3 $O/r05_v_pytest.txt
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2 | tee -a $O/r05_v_pytest.txt
timeout 900 python bench.py --save-traffic 2> $O/r05_v_c3.err | tail -1 > $O/r05_v_bench_c3.json; echo "default bench rc $?"
cp profiles/traffic_closest.json $O/r05_v_traffic_closest.json
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/r05_v_prof -o c3 --output-format csv -- python $R/bench.py --cpu-seconds 0 --traffic none --secondary off > $O/r05_v_bench_c3_under_rocprof.json 2> $O/r05_v_prof.err)
head -6 $O/r05_v_prof/c3_kernel_stats.csv | cut -c1-150
rm -f $O/r05_v_prof/c3_kernel_trace.csv
timeout 400 python bench.py --config c2 --cpu-seconds 10 --cpu-port-seconds 0 2> $O/r05_v_c2.err | tail -1 > $O/r05_v_bench_c2.json
timeout 600 python bench.py --config c4 --steps 2 --cpu-seconds 10 --cpu-port-seconds 0 2> $O/r05_v_c4.err | tail -1 > $O/r05_v_bench_c4.json
timeout 900 python bench.py --config c5 --steps 1 --warmup 1 --cpu-seconds 12 --cpu-port-seconds 0 2> $O/r05_v_c5.err | tail -1 > $O/r05_v_bench_c5.json
F="--steps 2 --warmup 1 --traffic none --cpu-port-seconds 0 --cpu-seconds 10"
timeout 500 python bench.py --subsurface $F 2> $O/r05_v_sss.err | tail -1 > $O/r05_v_bench_c3_subsurface.json
timeout 500 python bench.py --smokebox $F 2> $O/r05_v_smoke.err | tail -1 > $O/r05_v_bench_c3_smokebox.json
python - <<'EOF2'
import json
for c in ("c3", "c2", "c4", "c5", "c3_subsurface", "c3_smokebox"):
    try:
        d=json.load(open('/root/repo/gpurun_out/r05_v_bench_%s.json' % c)); r=d['roofline']
        print(c, d['value'], d['ms_per_step'], d['kernel_ms_per_step'], 'launch ms', round(r['avg_launch_ms'], 2), 'frac', r['frac'], 'alg8d', r.get('frac_alg_8d'), 'valu', r.get('frac_valu_lane_throughput'), r.get('nodes_per_ray'), (d.get('cpu_baseline') or {}).get('parity_crop', {}).get('pixels_within_tol'), (d.get('cpu_baseline') or {}).get('value'))
        if d.get('secondary'): print('   secondary', json.dumps(d['secondary'])[:700])
    except Exception as e: print(c, 'ERR', e)
EOF2
