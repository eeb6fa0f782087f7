#!/bin/bash
# round 3, GPU call s: alpha masks in wave-wide alpha phases (PT_ALPHA_DEFER).  Call r measured the C3 stand-in with masked leaves at 68.6 Msamples/s
# against 306 without masks while pbrt_ref on the host loses 14 %: the mask interpreter ran inside the leaf step for a lane or two at a time.
# Parity suite first (the emulator runs one lane per wave: this is the first multi-lane run of the phases), then the 16-spp probe frame for the
# shipped build (PT_ALPHA_MIN 16), thresholds 8 / 32 and the inline evaluation (PT_ALPHA_DEFER=0); in haze (walked shadow / MIS rays: TR instances)
# shipped vs inline; then the full frame with its pbrt_ref crop.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $O/r03_s_pytest.txt 2>&1; tail -3 $O/r03_s_pytest.txt
run() { tag=$1; shift; env "$@" 2> $O/r03_s_$tag.err | tail -1 > $O/r03_s_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r03_s_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], (d.get("cpu_baseline") or {}).get("parity_crop", {}).get("pixels_within_tol"), {k: round(v, 1) for k, v in t.items()} if isinstance(t, dict) else t)
except Exception as e: print("$tag", "ERR", e)
EOF2
}
V=$R/pbrt-v3-distributed_amd/lib/variants
P="--leafmask --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none"
run lm16_cur A=1 timeout 300 python bench.py $P
run lm16_inline PBRT_AMD_DEVICE_LIB=$V/adefer0.so timeout 300 python bench.py $P
run lm16_min8 PBRT_AMD_DEVICE_LIB=$V/amin8.so timeout 300 python bench.py $P
run lm16_min32 PBRT_AMD_DEVICE_LIB=$V/amin32.so timeout 300 python bench.py $P
run lmhaze16_cur A=1 timeout 300 python bench.py $P --volpath
run lmhaze16_inline PBRT_AMD_DEVICE_LIB=$V/adefer0.so timeout 300 python bench.py $P --volpath
run c3_leafmask A=1 timeout 500 python bench.py --leafmask --steps 2 --warmup 1 --cpu-seconds 10 --cpu-port-seconds 0 --traffic none
