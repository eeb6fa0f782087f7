#!/bin/bash
# round 5, GPU call f: (1) instruction-cache counters of k_shade on C3 (16 spp) in the three modes of call e -- is the gain of the class parts instruction-cache locality? --
# together with wave-state counters; (2) the textured + leaf-masked C3 with class parts (kinds of material; every part the generic textured instance) against one launch, twice
# each; (3) the headline at full size with the shipped configuration (class 2 = generic instance on its part), twice, against off.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
for mode in 1 generic 0; do
  export PBRT_AMD_SHADE_CLASSES=$mode
  (cd /tmp && timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_IFETCH -d $O/r05_f_pmc_$mode -o c --output-format csv -- python $R/bench.py --spp 16 --steps 1 --warmup 0 --cpu-seconds 0 --traffic none --pmc-child > /dev/null 2> $O/r05_f_pmc_$mode.err)
  python - <<EOF2
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob("$O/r05_f_pmc_$mode/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_shade" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void ", ""); agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, c in agg.items():
    print("mode $mode %-40s launches %2d  icache req %.4g hit %.4g miss %.4g dup %.4g (miss rate %.4f)  wave cycles %.4g wait_any %.3f wait_inst %.3f ifetch %.4g" % (k, len(n[k]), c["SQC_ICACHE_REQ"], c["SQC_ICACHE_HITS"], c["SQC_ICACHE_MISSES"], c["SQC_ICACHE_MISSES_DUPLICATE"], c["SQC_ICACHE_MISSES"] / max(1, c["SQC_ICACHE_REQ"]), c["SQ_WAVE_CYCLES"], c["SQ_WAIT_ANY"] / max(1, c["SQ_WAVE_CYCLES"]), c["SQ_WAIT_INST_ANY"] / max(1, c["SQ_WAVE_CYCLES"]), c["SQ_IFETCH"]))
EOF2
  rm -rf $O/r05_f_pmc_$mode
done
unset PBRT_AMD_SHADE_CLASSES
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $WHAT $BARGS --warmup 1 --cpu-port-seconds 0 --cpu-seconds 0 --traffic none 2> $O/r05_f_$tag.err | tail -1 > $O/r05_f_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r05_f_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()})
except Exception as e: print("$tag", "ERR", e)
EOF2
}
WHAT="--textured --leafmask"; BARGS="--spp 16 --steps 2"
run texlm_16_off_1 PBRT_AMD_SHADE_CLASSES=0
run texlm_16_parts_1 A=1
run texlm_16_off_2 PBRT_AMD_SHADE_CLASSES=0
run texlm_16_parts_2 A=1
WHAT=""; BARGS="--steps 3"
run c3_full_off_1 PBRT_AMD_SHADE_CLASSES=0
run c3_full_on_1 A=1
run c3_full_off_2 PBRT_AMD_SHADE_CLASSES=0
run c3_full_on_2 A=1
