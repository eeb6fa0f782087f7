#!/bin/bash
# round 3, GPU call r: alpha masks under volpath ride the segment walk (volTr = volWave && (hasNullMat || hasAlpha)).  Parity suite + smoke, then the
# C3 stand-in with its leaf quads as alpha-masked meshes (bench.py --leafmask: about half of the triangles carry a mask): under `path` at the quoted
# size with its pbrt_ref crop, and inside the haze under `volpath` at 16 spp -- walked (default) against the general form (PBRT_AMD_VOL_TR_QUEUES=0).
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
rm -f $O/r03_r_parity_report.jsonl
PBRT_AMD_PARITY_REPORT=$O/r03_r_parity_report.jsonl timeout 600 python -m pytest tests -m gpu -x -q > $O/r03_r_pytest.txt 2>&1; tail -3 $O/r03_r_pytest.txt
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2 | tee -a $O/r03_r_pytest.txt
run() { tag=$1; shift; env "$@" > $O/r03_r_bench_$tag.json.tmp 2> $O/r03_r_$tag.err; tail -1 $O/r03_r_bench_$tag.json.tmp > $O/r03_r_bench_$tag.json; rm -f $O/r03_r_bench_$tag.json.tmp
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r03_r_bench_$tag.json")); t = d.get("kernel_ms", {})
    print("$tag", d["value"], d["ms_per_step"], d.get("kernel_ms_per_step"), (d.get("cpu_baseline") or {}).get("parity_crop", {}).get("pixels_within_tol"), {k: round(v, 1) for k, v in t.items()} if isinstance(t, dict) else "")
except Exception as e: print("$tag", "ERR", e)
EOF2
}
run c3_leafmask_haze16 A=1 timeout 400 python bench.py --leafmask --volpath --spp 16 --steps 2 --warmup 1 --cpu-seconds 10 --cpu-port-seconds 0 --traffic none
run c3_leafmask_haze16_general PBRT_AMD_VOL_TR_QUEUES=0 timeout 400 python bench.py --leafmask --volpath --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none
run c3_leafmask A=1 timeout 500 python bench.py --leafmask --steps 2 --warmup 1 --cpu-seconds 10 --cpu-port-seconds 0 --traffic none
