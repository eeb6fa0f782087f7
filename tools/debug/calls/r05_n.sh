#!/bin/bash
# round 5, GPU call n: (1) per-lane lobe records with a four-word header instead of a zero-filled 100-byte record (AddLobe), (2) the alpha phase's thresholds on the MID shape
# (PT_ALPHA_MIN 16 / 32 / 48, PT_ALPHA_GO_MUL 2), (3) the probe walk's tail threshold with the hit list in place (PBRT_AMD_SSS_TAIL) -- 16 spp.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/pbrt-v3-distributed_amd/lib/variants
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $WHAT $BARGS --warmup 1 --cpu-port-seconds 0 --cpu-seconds 0 --traffic none --secondary off 2> $O/r05_n_$tag.err | tail -1 > $O/r05_n_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r05_n_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()})
except Exception as e: print("$tag", "ERR", e)
EOF2
}
BARGS="--spp 16 --steps 2"
WHAT="--textured --leafmask"; run texlm_default A=1; for v in amin16 amin48 ago2; do run texlm_$v PBRT_AMD_DEVICE_LIB=$V/$v.so; done
WHAT="--subsurface"; for t in 131072 262144 524288 1048576; do run sss_tail$t PBRT_AMD_SSS_TAIL=$t; done
