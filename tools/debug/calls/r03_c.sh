#!/bin/bash
# round 3, GPU call c: what bounds k_trace<QN> now?  Call b: 69 % of the node steps served from LDS (171.6 -> 69.5 vector requests per ray) bought
# 2.6 % in the 1024-thread shape, and that shape is 25 % slower than round 2's 6 x 256 threads with no hot nodes at all.  Here: hot nodes at 24 waves
# per CU (v768: 2 x 768 threads, 16 stack entries, K = 512; v256: 6 x 256 threads, 12 stack entries, K = 224) and the SQ counters of the cold shape
# and the current one side by side (VALU busy, SIMT efficiency, LDS conflicts, waits).
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
rocprofv3 -L 2>/dev/null | grep -oE "\bSQ_[A-Z_0-9]+" | sort -u | tr '\n' ' ' > $O/r03_c_sq_counters.txt; wc -w $O/r03_c_sq_counters.txt
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none 2>$O/r03_c_$tag.err | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('$tag', d['value'], d['kernel_ms_per_step'], 'hot share', r.get('hot_share_of_node_visits'), r.get('launch_shape'), 'req/ray', (r.get('request_rate') or {}).get('requests_per_ray'))" | tee -a $O/r03_c_ab_16spp.txt; }
for v in cold v768 v256; do run $v PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/$v.so; done
pmc() { tag=$1; lib=$2; shift; shift
  (cd /tmp && PBRT_AMD_DEVICE_LIB=$lib timeout 300 rocprofv3 --pmc "$@" -d $O/r03_c_pmc_$tag -o c --output-format csv -- python $R/bench.py --spp 8 --steps 1 --warmup 1 --cpu-seconds 0 --traffic none > /dev/null 2> $O/r03_c_pmc_$tag.log)
  python tools/profile_summary.py pmc $O/r03_c_pmc_$tag $O/r03_c_pmc_$tag.json > /dev/null 2>&1
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r03_c_pmc_$tag.json"))
    for k, v in d.items():
        if k.startswith("void k_trace<0, false") or k.startswith("void k_trace<2, false"): print("$tag", k[:22], {a: (b if a == "launches" else round(b / v["launches"])) for a, b in v.items()})
except Exception as e: print("pmc $tag:", e, open("$O/r03_c_pmc_$tag.log").read()[-300:])
EOF2
}
CUR=$R/pbrt-v3-distributed_amd/lib/libpbrt_amd.so; COLD=$R/pbrt-v3-distributed_amd/lib/variants/cold.so; V768=$R/pbrt-v3-distributed_amd/lib/variants/v768.so
for pair in "cold $COLD" "cur $CUR" "v768 $V768"; do set -- $pair
  pmc ${1}_valu $2 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
  pmc ${1}_mix $2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM
  pmc ${1}_wait $2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS
done
