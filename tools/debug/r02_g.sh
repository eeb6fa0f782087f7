#!/bin/bash
# round 2: what bounds the lean BVH4 traversal?  (1) occupancy sweep: 2 / 4 / 6 persistent blocks per CU, (2) L1->L2 requests and L2 hit rate with and without ray binning
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp; R=/root/repo
export PBRT_AMD_TRACE=bvh4
run() { timeout 200 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none 2>gpurun_out/r02g_$1.err | tee gpurun_out/r02g_$1.json | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('$1', d['value'], d['kernel_ms_per_step'])"; }
run grid6 | tee gpurun_out/r02g_ab.txt
PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/grid4.so run grid4 | tee -a gpurun_out/r02g_ab.txt
PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/grid2.so run grid2 | tee -a gpurun_out/r02g_ab.txt
pmc() { tag=$1; shift; (cd /tmp && timeout 300 rocprofv3 --pmc "$@" -d $R/gpurun_out/r02g_pmc_$tag -o c --output-format csv -- python $R/bench.py --spp 8 --steps 1 --warmup 0 --cpu-seconds 0 --traffic none --pmc-child > /dev/null 2> $R/gpurun_out/r02g_pmc_$tag.log); }
pmc bin TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum
PBRT_AMD_RAYBIN=0 pmc nobin TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum
pmc bin_sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS
python - <<'PY' | tee gpurun_out/r02g_pmc_summary.txt
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r02g_pmc_*/")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "k_trace" in k or "k_shade" in k or "k_raybin" in k:
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in agg:
        print(d, k, "launches", len(n[k]), {c: round(v / len(n[k])) for c, v in agg[k].items()})
PY
