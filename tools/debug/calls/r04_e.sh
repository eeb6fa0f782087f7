#!/bin/bash
# round 4, GPU call e: k_shade lost 17 % between two builds whose k_shade ISA is identical (call a: 37.7 ms per 16-spp frame, call d: 45.1) and got it back in a
# variant whose only difference is the cndmask encoding (larger code before and inside the kernel).  k_shade + its out-of-line callees are ~118 KB of code for a
# 64 KB instruction cache: is it code PLACEMENT?  Same compiler output, k_shade<false,0,false,false> shifted by .space N in front of it (tools/debug/asm_variant.sh).
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
rocprofv3 -L 2>/dev/null | grep -io 'SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*' | sort -u | tr '\n' ' ' > $O/r04_e_counters_icache.txt; cat $O/r04_e_counters_icache.txt; echo
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none 2> $O/r04_e_$tag.err | tail -1 > $O/r04_e_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r04_e_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()})
except Exception as e: print("$tag", "ERR", e)
EOF2
}
V=$R/pbrt-v3-distributed_amd/lib/variants
run shipped A=1
for v in pad0 pad8192 pad24576 pad40960 cnd64; do run $v PBRT_AMD_DEVICE_LIB=$V/$v.so; done
P="--spp 8 --steps 1 --warmup 1 --cpu-seconds 0 --traffic none"
for v in pad0 cnd64; do
(cd /tmp && PBRT_AMD_DEVICE_LIB=$V/$v.so timeout 200 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU -d $O/r04_e_pmc_$v -o c --output-format csv -- python $R/bench.py $P > /dev/null 2> $O/r04_e_pmc_$v.log)
python - <<EOF2
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob("$O/r04_e_pmc_$v/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:50]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); disp[k].add(row["Dispatch_Id"])
for k in sorted(agg):
    if "k_shade" in k or "k_trace<0, false" in k: print("$v", k, len(disp[k]), {a: "%.4g" % (b / len(disp[k])) for a, b in agg[k].items()})
EOF2
done
