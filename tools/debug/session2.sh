#!/bin/bash
# GPU session: parity tests, A/B of kernel variants (lib/variants/*.so), latency PMC pass
cd /root/repo; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/s2_pytest.txt
for v in "$@"; do
  PBRT_AMD_DEVICE_LIB=/root/repo/pbrt-v3-distributed_amd/lib/variants/$v.so timeout 300 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 2>gpurun_out/s2_$v.err | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('$v', d['value'], d['kernel_ms_per_step'], 'nodes/ray', round(r['nodes_per_ray'],2), 'tris/ray', round(r['tris_per_ray'],2))" | tee -a gpurun_out/s2_variants.txt
done
bash tools/debug/pmc.sh lat1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM
bash tools/debug/pmc.sh lat2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum
python - <<'PY'
import csv, glob, collections
for tag in ["lat1","lat2"]:
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("/root/repo/gpurun_out/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:40]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    for k, d in agg.items():
        print(tag, k, {c: "%.4g" % v for c, v in d.items()})
PY
