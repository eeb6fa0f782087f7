#!/bin/bash
# round 4, GPU call k: what two rays per lane would have to overcome.  Two register-resident rays per lane need ~40 more VGPRs = 4 instead of 6 waves per
# SIMD.  The SAME round-4 kernels (same instruction streams, 80 VGPRs) at 16 instead of 24 waves per CU: -DPT_TRACEQ_BLOCK=512 -> two 512-thread blocks per CU
# (16 LDS stack entries per lane + 512 hot nodes each, as shipped).  16-spp C3 frame, twice each, + SQ counters (8 spp).
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none 2> $O/r04_k_$tag.err | tail -1 > $O/r04_k_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r04_k_bench_$tag.json")); t = d.get("kernel_ms_per_step", {}); r = d["roofline"]
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()}, r.get("launch_shape"))
except Exception as e: print("$tag", "ERR", e)
EOF2
}
V=$R/pbrt-v3-distributed_amd/lib/variants
run w24_shipped A=1
run w16 PBRT_AMD_DEVICE_LIB=$V/w16.so
run w24_shipped_again A=1
run w16_again PBRT_AMD_DEVICE_LIB=$V/w16.so
P="--spp 8 --steps 1 --warmup 1 --cpu-seconds 0 --traffic none"
for v in w24 w16; do
  if [ $v = w16 ]; then export PBRT_AMD_DEVICE_LIB=$V/w16.so; else unset PBRT_AMD_DEVICE_LIB; fi
  (cd /tmp && timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/r04_k_pmc_$v -o c --output-format csv -- python $R/bench.py $P > /dev/null 2> $O/r04_k_pmc_$v.log)
  python - <<EOF2
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob("$O/r04_k_pmc_$v/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:50]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); disp[k].add(row["Dispatch_Id"])
for k in sorted(agg):
    if "k_trace<0, false" in k: print("$v", k, len(disp[k]), {a: "%.4g" % (b / len(disp[k])) for a, b in agg[k].items()})
EOF2
done
