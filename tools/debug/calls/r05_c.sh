#!/bin/bash
# round 5, GPU call c: the class instances of k_shade with the short-queue fallback (PathState::shade_cls_min): threshold sweep on C4 (maxdepth 30: many short bounces) and C3,
# the class instances at 2 waves per SIMD (256 VGPRs, no spills) and DynIter grains of 64 / 128 items against the default build (3 waves, grain 256); then the classes on / off pair at
# each BASELINE config's full size.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/pbrt-v3-distributed_amd/lib/variants
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $WHAT $BARGS --warmup 1 --cpu-port-seconds 0 --cpu-seconds 0 --traffic none 2> $O/r05_c_$tag.err | tail -1 > $O/r05_c_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r05_c_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()})
except Exception as e: print("$tag", "ERR", e)
EOF2
}
WHAT="--config c4"; BARGS="--spp 32 --steps 2"
run c4_32_off PBRT_AMD_SHADE_CLASSES=0
for m in 0 1000000 4000000 16000000; do run c4_32_min$m PBRT_AMD_SHADE_CLASS_MIN=$m; done
run c4_32_cls2 PBRT_AMD_DEVICE_LIB=$V/cls2.so
run c4_32_grain64 PBRT_AMD_DEVICE_LIB=$V/grain64.so
WHAT=""; BARGS="--spp 16 --steps 2"
run c3_16_off PBRT_AMD_SHADE_CLASSES=0
for m in 0 4000000 16000000; do run c3_16_min$m PBRT_AMD_SHADE_CLASS_MIN=$m; done
run c3_16_cls2 PBRT_AMD_DEVICE_LIB=$V/cls2.so
run c3_16_grain64 PBRT_AMD_DEVICE_LIB=$V/grain64.so
run c3_16_grain128 PBRT_AMD_DEVICE_LIB=$V/grain128.so
WHAT="--config c2"; BARGS="--spp 32 --steps 2"
run c2_32_default A=1
run c2_32_cls2 PBRT_AMD_DEVICE_LIB=$V/cls2.so
# full sizes, classes on / off
WHAT=""; BARGS="--steps 3"
run c3_full_off PBRT_AMD_SHADE_CLASSES=0
run c3_full_on A=1
WHAT="--config c2"; run c2_full_off PBRT_AMD_SHADE_CLASSES=0; run c2_full_on A=1
WHAT="--config c4"; BARGS="--steps 2"; run c4_full_off PBRT_AMD_SHADE_CLASSES=0; run c4_full_on A=1
