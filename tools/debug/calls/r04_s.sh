#!/bin/bash
# round 4, GPU call s (the last): MipEWA with its texel fetches in batches of four (csrc/pt_texture.h, same texels / weights / order of additions) -- the whole GPU suite on that
# build, the textured frame against the one-texel-at-a-time build (lib/variants/ewa1.so), the textured + masked line, and two counter passes that say what k_shade<..., TEX> waits for.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 700 python -m pytest tests -m gpu -x -q > $O/r04_s_pytest.txt 2>&1; tail -3 $O/r04_s_pytest.txt
run() { tag=$1; shift; env "$@" timeout 200 python bench.py $WHAT --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --cpu-port-seconds 0 --traffic none 2> $O/r04_s_$tag.err | tail -1 > $O/r04_s_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r04_s_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()})
except Exception as e: print("$tag", "ERR", e)
EOF2
}
V=$R/pbrt-v3-distributed_amd/lib/variants
WHAT=--textured
run tex_ewa4 A=1
run tex_ewa1 PBRT_AMD_DEVICE_LIB=$V/ewa1.so
WHAT="--textured --leafmask"
run texlm_ewa4 A=1
P="--textured --spp 8 --steps 1 --warmup 1 --cpu-seconds 0 --cpu-port-seconds 0 --traffic none"
pmc() { tag=$1; shift
  (cd /tmp && timeout 200 rocprofv3 --pmc "$@" -d $O/r04_s_pmc_$tag -o c --output-format csv -- python $R/bench.py $P > /dev/null 2> $O/r04_s_pmc_$tag.log)
  python - <<EOF2
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob("$O/r04_s_pmc_$tag/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:44]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); disp[k].add(row["Dispatch_Id"])
for k in sorted(agg):
    if "k_shade<" in k or "k_trace<0" in k: print("$tag", k, len(disp[k]), {a: "%.4g" % (b / len(disp[k])) for a, b in agg[k].items()})
EOF2
}
pmc waves SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pmc insts SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU
