#!/bin/bash
# round 5, GPU call aa: the spatial light pick's first links issued as soon as the hit point is known (PT_PICK_EARLY: 1 = voxel funcInt + guide word, 2 = the whole pick) --
# the GPU suite on the variant 2 library, then C3 at the quoted size (twice, alternating), C4 at 32 spp and the textured + masked C3 at 16 spp: shipped / 1 / 2.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/pbrt-v3-distributed_amd/lib/variants
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $WHAT $BARGS --warmup 1 --cpu-port-seconds 0 --cpu-seconds 0 --traffic none --secondary off 2> $O/r05_aa_$tag.err | tail -1 > $O/r05_aa_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r05_aa_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()})
except Exception as e: print("$tag", "ERR", e)
EOF2
}
PBRT_AMD_DEVICE_LIB=$V/pick2.so timeout 400 python -m pytest tests -m gpu -x -q -n 4 2>&1 | tail -1
PBRT_AMD_DEVICE_LIB=$V/pick1.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -n 4 -k "fixture or spatial or light" 2>&1 | tail -1
for rep in a b; do WHAT=""; BARGS="--steps 3"; run c3_full_${rep}_shipped A=1; run c3_full_${rep}_pick1 PBRT_AMD_DEVICE_LIB=$V/pick1.so; run c3_full_${rep}_pick2 PBRT_AMD_DEVICE_LIB=$V/pick2.so; done
WHAT="--config c4"; BARGS="--spp 32 --steps 2"; run c4_32_shipped A=1; run c4_32_pick1 PBRT_AMD_DEVICE_LIB=$V/pick1.so; run c4_32_pick2 PBRT_AMD_DEVICE_LIB=$V/pick2.so
WHAT="--textured --leafmask"; BARGS="--spp 16 --steps 2"; run texlm_16_shipped A=1; run texlm_16_pick1 PBRT_AMD_DEVICE_LIB=$V/pick1.so; run texlm_16_pick2 PBRT_AMD_DEVICE_LIB=$V/pick2.so
