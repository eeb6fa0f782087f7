#!/bin/bash
# round 3, GPU call f: pending-leaf traversal (a lane parks the leaf it reached and goes on with node steps; PT_PEND_LEAF) -- parity suite with the
# agreement figures written out (PBRT_AMD_PARITY_REPORT), A/B on the 16-spp C3 probe frame against the same build without it and with other
# phase lengths, SQ counters of the new kernel, the C3 line.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
rm -f $O/r03_f_parity_report.jsonl
PBRT_AMD_PARITY_REPORT=$O/r03_f_parity_report.jsonl timeout 900 python -m pytest tests -m gpu -x -q > $O/r03_f_pytest.txt 2>&1; tail -3 $O/r03_f_pytest.txt
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none 2>$O/r03_f_$tag.err | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('$tag', d['value'], d['kernel_ms_per_step'], 'nodes/ray', round(r['nodes_per_ray'],2), 'tris/ray', round(r['tris_per_ray'],2), 'hot', r.get('hot_share_of_node_visits'))" | tee -a $O/r03_f_ab_16spp.txt; }
run cur A=1
for v in nopend pend4 pend4_32; do run $v PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/$v.so; done
pmc() { tag=$1; lib=$2; shift; shift
  (cd /tmp && PBRT_AMD_DEVICE_LIB=$lib timeout 300 rocprofv3 --pmc "$@" -d $O/r03_f_pmc_$tag -o c --output-format csv -- python $R/bench.py --spp 8 --steps 1 --warmup 1 --cpu-seconds 0 --traffic none > /dev/null 2> $O/r03_f_pmc_$tag.log)
  python tools/profile_summary.py pmc $O/r03_f_pmc_$tag $O/r03_f_pmc_$tag.json > /dev/null 2>&1
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r03_f_pmc_$tag.json"))
    for k, v in d.items():
        if k.startswith("void k_trace<0, false") or k.startswith("void k_trace<2, false"): print("$tag", k[:22], {a: (b if a == "launches" else round(b / v["launches"])) for a, b in v.items()})
except Exception as e: print("pmc $tag:", e)
EOF2
}
pmc cur_valu $R/pbrt-v3-distributed_amd/lib/libpbrt_amd.so SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
pmc nopend_valu $R/pbrt-v3-distributed_amd/lib/variants/nopend.so SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
timeout 600 python bench.py --save-traffic 2> $O/r03_f_c3.err | tail -1 > $O/r03_f_bench_c3.json
cp profiles/traffic_closest.json $O/r03_f_traffic_closest.json
python - <<'EOF2'
import json
d=json.load(open('/root/repo/gpurun_out/r03_f_bench_c3.json')); r=d['roofline']
print('C3', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], 'launch ms', r['avg_launch_ms'], 'frac', r['frac'], (d.get('cpu_baseline') or {}).get('parity_crop'))
EOF2
python - <<'EOF2'
import json, collections
agg = collections.defaultdict(list)
for l in open('/root/repo/gpurun_out/r03_f_parity_report.jsonl'):
    r = json.loads(l); agg[r['test']].append(r)
for k, v in agg.items():
    keys = [x for x in v[0] if x.startswith('bit_identical') or x == 'within_tol']
    print(k, len(v), {x: (round(min(r[x] for r in v), 5), round(max(r[x] for r in v), 5)) for x in keys})
EOF2
