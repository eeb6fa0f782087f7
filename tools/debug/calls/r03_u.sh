#!/bin/bash
# round 3, GPU call u: the state after the alpha phases (PT_ALPHA_MIN 32) -- parity suite (incl. the masked-leaves reduced config) + smoke; the masked-leaves
# frame at the quoted size with its pbrt_ref crop and rocprofv3 kernel stats; the same in haze at 16 spp, walked vs general form; the default C3 line as a
# regression check of the mask-free instances.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
rm -f $O/r03_u_parity_report.jsonl
PBRT_AMD_PARITY_REPORT=$O/r03_u_parity_report.jsonl timeout 600 python -m pytest tests -m gpu -x -q > $O/r03_u_pytest.txt 2>&1; tail -3 $O/r03_u_pytest.txt
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2 | tee -a $O/r03_u_pytest.txt
run() { tag=$1; shift; env "$@" 2> $O/r03_u_$tag.err | tail -1 > $O/r03_u_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r03_u_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], (d.get("cpu_baseline") or {}).get("parity_crop", {}).get("pixels_within_tol"), {k: round(v, 1) for k, v in t.items()} if isinstance(t, dict) else t)
except Exception as e: print("$tag", "ERR", e)
EOF2
}
run c3_leafmask A=1 timeout 500 python bench.py --leafmask --cpu-seconds 10 --cpu-port-seconds 0 --traffic none
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/r03_u_prof -o lm --output-format csv -- python $R/bench.py --leafmask --steps 2 --warmup 1 --cpu-seconds 0 --traffic none > $O/r03_u_bench_c3_leafmask_under_rocprof.json 2> $O/r03_u_prof.err)
head -6 $O/r03_u_prof/lm_kernel_stats.csv | cut -c1-170
P="--leafmask --volpath --spp 16 --steps 2 --warmup 1 --traffic none"
run lmhaze16_walked A=1 timeout 300 python bench.py $P --cpu-seconds 10 --cpu-port-seconds 0
run lmhaze16_general PBRT_AMD_VOL_TR_QUEUES=0 timeout 300 python bench.py $P --cpu-seconds 0
run c3 A=1 timeout 400 python bench.py --cpu-seconds 0 --traffic none
