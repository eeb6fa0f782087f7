#!/bin/bash
# round 4, GPU call g: the interior step with the cheaper tail (TravNodeStepQ2 / TravStackB: sign-bit hit decision, miss-folded integer keys, unconditional pushes,
# speculative pop, per-wave deep-stack branch) + v_max3 |.| in the triangle test's error bounds -- parity suite, then A/B against -DPT_STEP2=0 on the 16-spp C3 frame
# (twice each, interleaved), then SQ counters of both (8 spp).
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/r04_g_pytest.txt 2>&1; tail -3 $O/r04_g_pytest.txt
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none 2> $O/r04_g_$tag.err | tail -1 > $O/r04_g_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r04_g_bench_$tag.json")); t = d.get("kernel_ms_per_step", {}); r = d["roofline"]
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()}, "nodes/ray", round(r["nodes_per_ray"], 2), "tris/ray", round(r["tris_per_ray"], 2))
except Exception as e: print("$tag", "ERR", e)
EOF2
}
V=$R/pbrt-v3-distributed_amd/lib/variants
run step2 A=1
run step1 PBRT_AMD_DEVICE_LIB=$V/step1.so
run step2_again A=1
run step2_third A=1
run step1_again PBRT_AMD_DEVICE_LIB=$V/step1.so
P="--spp 8 --steps 1 --warmup 1 --cpu-seconds 0 --traffic none"
for v in step2 step1; do
  if [ $v = step1 ]; then export PBRT_AMD_DEVICE_LIB=$V/step1.so; else unset PBRT_AMD_DEVICE_LIB; fi
  (cd /tmp && timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $O/r04_g_pmc_$v -o c --output-format csv -- python $R/bench.py $P > /dev/null 2> $O/r04_g_pmc_$v.log)
  python - <<EOF2
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob("$O/r04_g_pmc_$v/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:50]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); disp[k].add(row["Dispatch_Id"])
for k in sorted(agg):
    if "k_trace<0, false" in k or "k_trace<2, false" in k: print("$v", k, len(disp[k]), {a: "%.4g" % (b / len(disp[k])) for a, b in agg[k].items()})
EOF2
done
