#!/bin/bash
# round 5, GPU call h: with the interleaved partition in place (one launch, PBRT_AMD_SHADE_CLASSES=0), k_shade's occupancy and grain re-measured: 2 / 3 / 4 waves per SIMD
# (256 / 168 / 128 VGPRs) and DynIter grains of 64 / 128 / 256 / 512 items -- C3 at 16 spp, C2 at 32 spp, C4 at 32 spp, each variant twice (alternating).
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/pbrt-v3-distributed_amd/lib/variants
export PBRT_AMD_SHADE_CLASSES=0
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $WHAT $BARGS --warmup 1 --cpu-port-seconds 0 --cpu-seconds 0 --traffic none 2> $O/r05_h_$tag.err | tail -1 > $O/r05_h_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r05_h_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()})
except Exception as e: print("$tag", "ERR", e)
EOF2
}
for rep in a b; do
WHAT=""; BARGS="--spp 16 --steps 3"
run c3_16_${rep}_default A=1
for v in w2 w4 g64 g128 g512; do run c3_16_${rep}_$v PBRT_AMD_DEVICE_LIB=$V/$v.so; done
done
WHAT="--config c2"; BARGS="--spp 32 --steps 3"
run c2_32_default A=1
for v in w2 w4 g64 g128 g512; do run c2_32_$v PBRT_AMD_DEVICE_LIB=$V/$v.so; done
WHAT="--config c4"; BARGS="--spp 32 --steps 2"
run c4_32_default A=1
for v in w2 w4 g64 g128 g512; do run c4_32_$v PBRT_AMD_DEVICE_LIB=$V/$v.so; done
