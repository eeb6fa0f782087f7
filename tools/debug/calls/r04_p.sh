#!/bin/bash
# round 4, GPU call p: k_shade with the next item's PathRec line and triangle records touched one group ahead (PT_SHADE_PREFETCH, csrc/pbrt_amd.hip) against the
# same source without (lib/variants/nopf.so), C3 at 16 spp, alternating; parity of the shipped build; the subsurface line with the untextured k_shade for the
# plain materials, and the tail threshold of the probe walk at 65536 (shipped) / 131072 / 262144.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
run() { tag=$1; shift; env "$@" timeout 200 python bench.py $WHAT --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none 2> $O/r04_p_$tag.err | tail -1 > $O/r04_p_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r04_p_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()})
except Exception as e: print("$tag", "ERR", e)
EOF2
}
V=$R/pbrt-v3-distributed_amd/lib/variants
WHAT=
run prefetch_1 A=1
run nopf_1 PBRT_AMD_DEVICE_LIB=$V/nopf.so
run prefetch_2 A=1
run nopf_2 PBRT_AMD_DEVICE_LIB=$V/nopf.so
timeout 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "baseline_configs_reduced or fixture or li_per_sample or only_the_vertices" 2>&1 | tail -2
WHAT=--subsurface
run sss_tail65536 A=1
run sss_tail131072 PBRT_AMD_SSS_TAIL=131072
run sss_tail262144 PBRT_AMD_SSS_TAIL=262144
WHAT=--textured
run tex_prefetch A=1
run tex_nopf PBRT_AMD_DEVICE_LIB=$V/nopf.so
