#!/bin/bash
# round 5, GPU call j: the texture routines without per-call copies (TexCtx by reference, constant nodes answered in line, (u, v) mapping in line, no zero-filled value array),
# wave-uniform tables on by default (texuni0 = lib/variants/texuni0.so = PT_TEX_UNIFORM=0) -- textured + leaf-masked, textured only, leaf-masked only C3 at 16 spp, and the
# texture parity tests on the hardware.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/pbrt-v3-distributed_amd/lib/variants
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -n 4 -k "tex or alpha or material or fixture" 2>&1 | tail -2 | tee $O/r05_j_pytest_tex.txt
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $WHAT $BARGS --warmup 1 --cpu-port-seconds 0 --cpu-seconds 0 --traffic none 2> $O/r05_j_$tag.err | tail -1 > $O/r05_j_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r05_j_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()})
except Exception as e: print("$tag", "ERR", e)
EOF2
}
BARGS="--spp 16 --steps 2"
WHAT="--textured --leafmask"; run texlm_default A=1; run texlm_texuni0 PBRT_AMD_DEVICE_LIB=$V/texuni0.so
WHAT="--textured"; run tex_default A=1
WHAT="--leafmask"; run lm_default A=1
