#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
bash tools/debug/pmc.sh e1 SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_IFETCH
bash tools/debug/pmc.sh e2 SQC_ICACHE_MISSES SQC_ICACHE_HITS SQC_ICACHE_REQ SQC_DCACHE_MISSES SQC_DCACHE_HITS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64
bash tools/debug/pmc.sh e3 SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32
python - <<'PY'
import csv, glob, collections
for tag in ["e1","e2","e3"]:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob("/root/repo/gpurun_out/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:40]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in agg.items():
        if k.startswith("k_shade") or "k_trace<0, false" in k: print(tag, k, {c: "%.4g" % v for c, v in d.items()})
PY
