#!/bin/bash
# round 4, GPU call d: does v_cndmask_b32_e32 (VCC implicit) really cost 23 cycles per wave instruction (tools/valu_probe/issue_probe, r04_b/c: 22.9 vs 4.1 for
# the _e64 encoding) in the REAL kernels?  A/B of the shipped library against lib/variants/cnd64.so = the same compiler output with every
# `v_cndmask_b32_e32 ..., vcc` re-encoded as `v_cndmask_b32_e64 ..., vcc` (tools/debug/asm_variant.sh), 16-spp C3 frame, C4 (shade-bound) at 64 spp.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none 2> $O/r04_d_$tag.err | tail -1 > $O/r04_d_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r04_d_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()})
except Exception as e: print("$tag", "ERR", e)
EOF2
}
V=$R/pbrt-v3-distributed_amd/lib/variants
run shipped A=1
run cnd64 PBRT_AMD_DEVICE_LIB=$V/cnd64.so
run shipped_again A=1
run cnd64_again PBRT_AMD_DEVICE_LIB=$V/cnd64.so
PBRT_AMD_DEVICE_LIB=$V/cnd64.so timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "closest_hit or render_vs_reference or li_per_sample" 2>&1 | tail -2
