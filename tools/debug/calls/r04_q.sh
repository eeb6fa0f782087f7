#!/bin/bash
# round 4, GPU call q: the state the round ends with (after the rcp boxes, the probe-walk tail kernel and the routed subsurface shading) -- parity suite + smoke, every BASELINE config at its quoted size with its pbrt_ref crop and the live roofline
# (C3 = the default line: live FETCH_SIZE + SQ passes, measured shader clock, the three named fractions; C2, C4, C5), rocprofv3 kernel stats of the default command,
# the secondary lines (textured + leaf masks combined, subsurface, smoke box).  (The N > 1 path did not change after call i: profiles/r04_i_n2_one_device_gloo.txt, r04_i_scale_model_c3.json.)
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
rm -f $O/r04_q_parity_report.jsonl
PBRT_AMD_PARITY_REPORT=$O/r04_q_parity_report.jsonl timeout 900 python -m pytest tests -m gpu -x -q > $O/r04_q_pytest.txt 2>&1; tail -3 $O/r04_q_pytest.txt
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2 | tee -a $O/r04_q_pytest.txt
timeout 700 python bench.py --save-traffic 2> $O/r04_q_c3.err | tail -1 > $O/r04_q_bench_c3.json
cp profiles/traffic_closest.json $O/r04_q_traffic_closest.json
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/r04_q_prof -o c3 --output-format csv -- python $R/bench.py --cpu-seconds 0 --traffic none > $O/r04_q_bench_c3_under_rocprof.json 2> $O/r04_q_prof.err)
head -5 $O/r04_q_prof/c3_kernel_stats.csv | cut -c1-150
timeout 400 python bench.py --config c2 --cpu-seconds 10 --cpu-port-seconds 0 2> $O/r04_q_c2.err | tail -1 > $O/r04_q_bench_c2.json
timeout 600 python bench.py --config c4 --steps 2 --cpu-seconds 10 --cpu-port-seconds 0 2> $O/r04_q_c4.err | tail -1 > $O/r04_q_bench_c4.json
timeout 900 python bench.py --config c5 --steps 1 --warmup 1 --cpu-seconds 12 --cpu-port-seconds 0 2> $O/r04_q_c5.err | tail -1 > $O/r04_q_bench_c5.json
F="--steps 2 --warmup 1 --traffic none --cpu-port-seconds 0 --cpu-seconds 10"
timeout 600 python bench.py --textured --leafmask $F 2> $O/r04_q_texlm.err | tail -1 > $O/r04_q_bench_c3_textured_leafmask.json
timeout 500 python bench.py --subsurface $F 2> $O/r04_q_sss.err | tail -1 > $O/r04_q_bench_c3_subsurface.json
timeout 500 python bench.py --smokebox $F 2> $O/r04_q_smoke.err | tail -1 > $O/r04_q_bench_c3_smokebox.json
python - <<'EOF2'
import json
for c in ("c3", "c2", "c4", "c5", "c3_textured_leafmask", "c3_subsurface", "c3_smokebox"):
    try:
        d=json.load(open('/root/repo/gpurun_out/r04_q_bench_%s.json' % c)); r=d['roofline']
        print(c, d['value'], d['ms_per_step'], d['kernel_ms_per_step'], 'launch ms', round(r['avg_launch_ms'], 2), 'frac', r['frac'], 'alg8d', r.get('frac_alg_8d'), 'valu', r.get('frac_valu_lane_throughput'), (r.get('valu_issue') or {}), r.get('shader_clock_GHz', {}).get('closest'), (d.get('cpu_baseline') or {}).get('parity_crop', {}).get('pixels_within_tol'), (d.get('cpu_baseline') or {}).get('value'))
    except Exception as e: print(c, 'ERR', e)
EOF2
