#!/bin/bash
# round 3, GPU call b: hot BVH4Q nodes in LDS (1024-thread traversal blocks) + the light-selection guide table.
# Parity suite first, then A/Bs on the 16-spp C3 probe frame: the built library, the same with PBRT_AMD_HOT=0 (big blocks, no hot nodes) and with
# PBRT_AMD_LIGHT_GUIDE=0, and the variant builds (cold = round 2's launch shape, hot1536 = 16 stack entries + 1536 nodes, hot512 = 2 x 512-thread blocks);
# then the full C3 line with live roofline, its rocprofv3 kernel stats and SQ / LDS counters of the shipped traversal kernel.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/r03_b_pytest.txt 2>&1; tail -3 $O/r03_b_pytest.txt
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none 2>$O/r03_b_$tag.err | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('$tag', d['value'], d['kernel_ms_per_step'], 'hot share', r.get('hot_share_of_node_visits'), r.get('launch_shape'), 'req/ray', (r.get('request_rate') or {}).get('requests_per_ray'))" | tee -a $O/r03_b_ab_16spp.txt; }
run cur A=1
run cur_hot0 PBRT_AMD_HOT=0
run cur_guide0 PBRT_AMD_LIGHT_GUIDE=0
for v in cold hot1536 hot512; do run $v PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/$v.so; done
timeout 600 python bench.py --save-traffic 2> $O/r03_b_c3.err | tail -1 > $O/r03_b_bench_c3.json; head -c 400 $O/r03_b_bench_c3.json; echo
cp profiles/traffic_closest.json $O/r03_b_traffic_closest.json
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/r03_b_prof -o c3 --output-format csv -- python $R/bench.py --cpu-seconds 0 --traffic none > $O/r03_b_bench_c3_under_rocprof.json 2> $O/r03_b_prof.err)
head -8 $O/r03_b_prof/c3_kernel_stats.csv | cut -c1-160
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY"; do
  tag=$(echo $set | cut -d' ' -f2)
  (cd /tmp && timeout 300 rocprofv3 --pmc $set -d $O/r03_b_pmc_$tag -o c --output-format csv -- python $R/bench.py --spp 8 --steps 1 --warmup 1 --cpu-seconds 0 --traffic none > /dev/null 2> $O/r03_b_pmc_$tag.log)
  python tools/profile_summary.py pmc $O/r03_b_pmc_$tag $O/r03_b_pmc_$tag.json > /dev/null 2>&1
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r03_b_pmc_$tag.json"))
    for k, v in d.items():
        if k.startswith("void k_trace<0, false") or k.startswith("void k_trace<2, false") or k.startswith("void k_shade<"): print(k[:60], {a: (b if a == "launches" else round(b / v["launches"])) for a, b in v.items()})
except Exception as e: print("pmc $tag:", e)
EOF2
done
python - <<'EOF'
import json
d=json.load(open('/root/repo/gpurun_out/r03_b_bench_c3.json')); r=d['roofline']
print('C3', d['value'], d['ms_per_step'], d['kernel_ms_per_step']); print({k: r[k] for k in r if k not in ('request_rate',)}); print(r.get('request_rate'))
print((d.get('cpu_baseline') or {}).get('parity_crop'))
EOF
