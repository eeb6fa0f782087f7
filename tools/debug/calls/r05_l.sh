#!/bin/bash
# round 5, GPU call l: what the mask evaluations and the per-lane material evaluation COST, by doing each twice (lib/variants/alpha2.so = PT_ALPHA_TWICE, mat2.so = PT_MAT_TWICE;
# measurement builds, wrong by construction only in time) -- leaf-masked and textured C3 at 16 spp.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/pbrt-v3-distributed_amd/lib/variants
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $WHAT $BARGS --warmup 1 --cpu-port-seconds 0 --cpu-seconds 0 --traffic none 2> $O/r05_l_$tag.err | tail -1 > $O/r05_l_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r05_l_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()})
except Exception as e: print("$tag", "ERR", e)
EOF2
}
BARGS="--spp 16 --steps 2"
WHAT="--leafmask"; run lm_default A=1; run lm_alpha2 PBRT_AMD_DEVICE_LIB=$V/alpha2.so
WHAT="--textured"; run tex_default A=1; run tex_mat2 PBRT_AMD_DEVICE_LIB=$V/mat2.so
