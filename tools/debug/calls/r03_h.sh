#!/bin/bash
# round 3, GPU call h: batched result hand-over (PT_BATCH_FINALIZE) A/B on the 16-spp C3 probe frame, the parity suite on the final kernels, and the
# final lines of the round: C3 (+ rocprofv3 kernel stats of the same command), C5 (4K / 512 spp) with its pbrt_ref crop.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/r03_h_pytest.txt 2>&1; tail -3 $O/r03_h_pytest.txt
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2 | tee -a $O/r03_h_pytest.txt
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --spp 16 --steps 3 --warmup 1 --cpu-seconds 0 --traffic none 2>$O/r03_h_$tag.err | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('$tag', d['value'], d['kernel_ms_per_step'])" | tee -a $O/r03_h_ab_16spp.txt; }
run cur A=1
run nobatch PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/nobatch.so
run cur_again A=1
timeout 600 python bench.py --save-traffic 2> $O/r03_h_c3.err | tail -1 > $O/r03_h_bench_c3.json
cp profiles/traffic_closest.json $O/r03_h_traffic_closest.json
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/r03_h_prof -o c3 --output-format csv -- python $R/bench.py --cpu-seconds 0 --traffic none > $O/r03_h_bench_c3_under_rocprof.json 2> $O/r03_h_prof.err)
head -6 $O/r03_h_prof/c3_kernel_stats.csv | cut -c1-150
timeout 900 python bench.py --config c5 --steps 1 --warmup 1 --cpu-seconds 30 2> $O/r03_h_c5.err | tail -1 > $O/r03_h_bench_c5.json
python - <<'EOF2'
import json
for c in ("c3", "c5"):
    try:
        d=json.load(open('/root/repo/gpurun_out/r03_h_bench_%s.json' % c)); r=d['roofline']
        print(c, d['value'], d['ms_per_step'], d['kernel_ms_per_step'], 'launch ms', r['avg_launch_ms'], 'frac', r['frac'], (d.get('cpu_baseline') or {}).get('parity_crop'))
    except Exception as e: print(c, 'ERR', e)
EOF2
