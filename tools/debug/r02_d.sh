#!/bin/bash
# round 2: GPU suite; PMC passes on the lean traversal kernel (address translation / L1 / TA counters); the bench line with the same-crop parity
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp; R=/root/repo
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee gpurun_out/r02d_pytest.txt
pmc() { tag=$1; shift; (cd /tmp && timeout 300 rocprofv3 --pmc "$@" -d $R/gpurun_out/r02d_pmc_$tag -o c --output-format csv -- python $R/bench.py --spp 8 --steps 1 --warmup 0 --cpu-seconds 0 --traffic none --pmc-child > /dev/null 2> $R/gpurun_out/r02d_pmc_$tag.log); }
pmc tlb TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_PERMISSION_MISS_sum
pmc tcp TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
pmc tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum
pmc ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
pmc sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pmc grbm GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY
python - <<'PY' | tee gpurun_out/r02d_pmc_summary.txt
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r02d_pmc_*/")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "k_trace" in k or "k_shade" in k:
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in agg:
        print(d, k, "launches", len(n[k]), {c: round(v / len(n[k])) for c, v in agg[k].items()})
PY
timeout 900 python bench.py --cpu-port-seconds 0 > gpurun_out/r02d_bench.json 2> gpurun_out/r02d_bench.err; tail -c 2500 gpurun_out/r02d_bench.json; tail -3 gpurun_out/r02d_bench.err
