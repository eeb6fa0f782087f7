#!/bin/bash
# round 5, GPU call u: k_shade's persistent grid -- 12 blocks per CU (four rounds of what is resident at 3 waves per SIMD; sized for round 1's static partition) against 6 and 3
# (one round: DynIter hands the work out dynamically) -- C4 at 32 spp (many short bounces: launch floor), C3 at 16 spp, C2 at 32 spp, textured C3 at 16 spp, each twice.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/pbrt-v3-distributed_amd/lib/variants
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $WHAT $BARGS --warmup 1 --cpu-port-seconds 0 --cpu-seconds 0 --traffic none --secondary off 2> $O/r05_u_$tag.err | tail -1 > $O/r05_u_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r05_u_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()})
except Exception as e: print("$tag", "ERR", e)
EOF2
}
for rep in a b; do
WHAT="--config c4"; BARGS="--spp 32 --steps 2"; run c4_32_${rep}_g12 A=1; run c4_32_${rep}_g6 PBRT_AMD_DEVICE_LIB=$V/sg6.so; run c4_32_${rep}_g3 PBRT_AMD_DEVICE_LIB=$V/sg3.so
WHAT=""; BARGS="--spp 16 --steps 3"; run c3_16_${rep}_g12 A=1; run c3_16_${rep}_g6 PBRT_AMD_DEVICE_LIB=$V/sg6.so; run c3_16_${rep}_g3 PBRT_AMD_DEVICE_LIB=$V/sg3.so
done
WHAT="--config c2"; BARGS="--spp 32 --steps 3"; run c2_32_g12 A=1; run c2_32_g6 PBRT_AMD_DEVICE_LIB=$V/sg6.so; run c2_32_g3 PBRT_AMD_DEVICE_LIB=$V/sg3.so
WHAT="--textured"; BARGS="--spp 16 --steps 2"; run tex_16_g12 A=1; run tex_16_g6 PBRT_AMD_DEVICE_LIB=$V/sg6.so; run tex_16_g3 PBRT_AMD_DEVICE_LIB=$V/sg3.so
