#!/bin/bash
# round 3, GPU call d: the final traversal shape (2 x 768 threads per CU, 16 LDS stack entries + 512 hot nodes per block) with the lean node-step
# tail (v_min / v_max compare-exchanges, one-address pushes): parity suite, A/B on the 16-spp C3 probe frame against the same build without the lean
# tail, with packed FMAs, and against round 2's shape (PT_HOT_NODES=0) with / without the lean tail; then the C3 line with live roofline + rocprofv3
# kernel stats, and the C2 / C4 lines on the final kernels.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/r03_d_pytest.txt 2>&1; tail -3 $O/r03_d_pytest.txt
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none 2>$O/r03_d_$tag.err | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('$tag', d['value'], d['kernel_ms_per_step'], 'hot share', r.get('hot_share_of_node_visits'), r.get('launch_shape'))" | tee -a $O/r03_d_ab_16spp.txt; }
run cur A=1
for v in nolean pkfma coldlean coldnolean; do run $v PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/$v.so; done
timeout 600 python bench.py --save-traffic 2> $O/r03_d_c3.err | tail -1 > $O/r03_d_bench_c3.json; head -c 300 $O/r03_d_bench_c3.json; echo
cp profiles/traffic_closest.json $O/r03_d_traffic_closest.json
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/r03_d_prof -o c3 --output-format csv -- python $R/bench.py --cpu-seconds 0 --traffic none > $O/r03_d_bench_c3_under_rocprof.json 2> $O/r03_d_prof.err)
head -6 $O/r03_d_prof/c3_kernel_stats.csv | cut -c1-150
timeout 400 python bench.py --config c2 --cpu-seconds 8 --cpu-port-seconds 0 2> $O/r03_d_c2.err | tail -1 > $O/r03_d_bench_c2.json
timeout 600 python bench.py --config c4 --steps 2 --cpu-seconds 8 --cpu-port-seconds 0 2> $O/r03_d_c4.err | tail -1 > $O/r03_d_bench_c4.json
python - <<'EOF2'
import json,glob
for f in sorted(glob.glob('/root/repo/gpurun_out/r03_d_bench_c*.json')):
    try:
        d=json.load(open(f)); r=d.get('roofline') or {}; c=d.get('cpu_baseline') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'frac', r.get('frac'), 'launch ms', r.get('avg_launch_ms'), 'hot', r.get('hot_share_of_node_visits'), 'crop', (c.get('parity_crop') or {}).get('pixels_within_tol'))
        print('   kernels', d.get('kernel_ms_per_step'))
    except Exception as e: print(f, 'ERR', e)
EOF2
