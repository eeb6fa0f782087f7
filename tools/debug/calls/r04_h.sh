#!/bin/bash
# round 4, GPU call h: k_shade with the DevScene behind a pointer that is laundered once per item (SGPR spills 172 -> 113, scratch 96 -> 48 B) against the
# shipped by-value form -- 16-spp C3 frame and the shade-bound C4 frame at 32 spp; the parity tests that exercise k_shade on the variant.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
run() { tag=$1; shift; env "$@" timeout 300 python bench.py $BARGS --steps 2 --warmup 1 --cpu-seconds 0 --traffic none 2> $O/r04_h_$tag.err | tail -1 > $O/r04_h_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r04_h_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()})
except Exception as e: print("$tag", "ERR", e)
EOF2
}
V=$R/pbrt-v3-distributed_amd/lib/variants
BARGS="--spp 16"
run c3_shipped A=1
run c3_argptr PBRT_AMD_DEVICE_LIB=$V/argptr.so
run c3_shipped_again A=1
run c3_argptr_again PBRT_AMD_DEVICE_LIB=$V/argptr.so
BARGS="--config c4 --spp 32"
run c4_shipped A=1
run c4_argptr PBRT_AMD_DEVICE_LIB=$V/argptr.so
PBRT_AMD_DEVICE_LIB=$V/argptr.so timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "render_vs_reference or li_per_sample or baseline_configs or many_lights" 2>&1 | tail -2
