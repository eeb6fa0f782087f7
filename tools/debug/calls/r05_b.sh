#!/bin/bash
# round 5, GPU call b: material-class instances of k_shade (k_shade<..., CLS>: matte / diffuse+glossy / specular-only parts of the material-sorted queue, everything in line) --
# (1) bit-identity of the films against the one generic launch on the GPU (tools/debug/cls_check.py), (2) A/B at 16 spp on C3, and on C2 / C4 at 32 spp: classes off, and the
# class instances at 3 / 4 / 5 waves per SIMD (168 / 128 / 96 VGPRs), (3) the L2's memory-side requests of the closest-hit kernel BY SIZE (TCC_EA0_RDREQ_64B / _128B) -- exact bytes
# instead of FETCH_SIZE's 64 B per request -- on the calibration probes and on the C3 workload.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/pbrt-v3-distributed_amd/lib/variants
timeout 300 python tools/debug/cls_check.py > $O/r05_b_cls_check.txt 2>&1; tail -6 $O/r05_b_cls_check.txt
run() { tag=$1; shift; env "$@" timeout 400 python bench.py $WHAT $BARGS --steps 2 --warmup 1 --cpu-port-seconds 0 --cpu-seconds 0 --traffic none 2> $O/r05_b_$tag.err | tail -1 > $O/r05_b_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r05_b_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()})
except Exception as e: print("$tag", "ERR", e)
EOF2
}
WHAT=""; BARGS="--spp 16"
run c3_16_off PBRT_AMD_SHADE_CLASSES=0
run c3_16_cls4 A=1
run c3_16_cls3 PBRT_AMD_DEVICE_LIB=$V/cls3.so
run c3_16_cls5 PBRT_AMD_DEVICE_LIB=$V/cls5.so
WHAT="--config c2"; BARGS="--spp 32"
run c2_32_off PBRT_AMD_SHADE_CLASSES=0
run c2_32_cls4 A=1
run c2_32_cls3 PBRT_AMD_DEVICE_LIB=$V/cls3.so
run c2_32_cls5 PBRT_AMD_DEVICE_LIB=$V/cls5.so
WHAT="--config c4"; BARGS="--spp 32"
run c4_32_off PBRT_AMD_SHADE_CLASSES=0
run c4_32_cls4 A=1
run c4_32_cls3 PBRT_AMD_DEVICE_LIB=$V/cls3.so
run c4_32_cls5 PBRT_AMD_DEVICE_LIB=$V/cls5.so
# (3) request sizes
(cd /tmp && timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_32B_sum -d $O/r05_b_calib_sz -o c --output-format csv -- python $R/tools/debug/fetch_calib.py 8 > $O/r05_b_calib_sz.txt 2>&1)
python tools/debug/fetch_calib_summary.py $O/r05_b_calib_sz 8 | tee $O/r05_b_calib_sz_summary.txt
(cd /tmp && timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum -d $O/r05_b_c3_sz -o c --output-format csv -- python $R/bench.py --spp 16 --steps 1 --warmup 0 --cpu-seconds 0 --traffic none --pmc-child > $O/r05_b_c3_sz.txt 2>&1)
python - <<'EOF2' | tee $O/r05_b_c3_sz_summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob('/root/repo/gpurun_out/r05_b_c3_sz/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
for k in sorted(agg, key=lambda k: -agg[k].get('TCC_EA0_RDREQ_sum', 0))[:8]:
    c = agg[k]; n = len(disp[k])
    b = 64 * c.get('TCC_EA0_RDREQ_64B_sum', 0) + 128 * c.get('TCC_EA0_RDREQ_128B_sum', 0)
    print('%-70s launches %3d  RDREQ %.4g  64B %.4g  128B %.4g  DRAM %.4g  per launch: %.3f GB by size (FETCH_SIZE-style 64 B x RDREQ: %.3f GB)' % (k[:70], n, c.get('TCC_EA0_RDREQ_sum', 0), c.get('TCC_EA0_RDREQ_64B_sum', 0), c.get('TCC_EA0_RDREQ_128B_sum', 0), c.get('TCC_EA0_RDREQ_DRAM_sum', 0), b / n / 1e9, 64 * c.get('TCC_EA0_RDREQ_sum', 0) / n / 1e9))
EOF2
find $O/r05_b_calib_sz $O/r05_b_c3_sz -name "*.csv" -size +2M -delete
