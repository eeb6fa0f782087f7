#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
for v in "$@"; do
  PBRT_AMD_DEVICE_LIB=/root/repo/pbrt-v3-distributed_amd/lib/variants/$v.so timeout 300 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 2>gpurun_out/s3_$v.err | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('$v', d['value'], d['kernel_ms_per_step'], 'nodes/ray', round(r['nodes_per_ray'],2), 'tris/ray', round(r['tris_per_ray'],2))" | tee -a gpurun_out/s3_variants.txt
done
bash tools/debug/pmc.sh d1 VmemLatency
bash tools/debug/pmc.sh d2 MeanOccupancyPerActiveCU VALUBusy
bash tools/debug/pmc.sh d3 MemUnitStalled SQ_WAIT_ANY SQ_WAVE_CYCLES
bash tools/debug/pmc.sh d4 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum
python - <<'PY'
import csv, glob, collections
for tag in ["d1","d2","d3","d4"]:
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for f in glob.glob("/root/repo/gpurun_out/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:40]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    for k, d in agg.items():
        if k.startswith("__amd"): continue
        print(tag, k, {c: "sum %.4g mean %.4g" % (v, v / n[k][c]) for c, v in d.items()})
PY
