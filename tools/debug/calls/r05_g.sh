#!/bin/bash
# round 5, GPU call g: DynIter's eighths interleaved grain by grain (every XCD shades the same mix of materials) against the contiguous eighths (lib/variants/contig.so), with
# one launch (PBRT_AMD_SHADE_CLASSES=0) and with the class parts -- C3 at 16 spp twice, then full size; C2, C4 at full size; the textured + leaf-masked, subsurface and smoke-box
# variants of C3 at 16 spp (k_shade<TEX>, k_shade_vol take their items through the same iterator).
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/pbrt-v3-distributed_amd/lib/variants
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $WHAT $BARGS --warmup 1 --cpu-port-seconds 0 --cpu-seconds 0 --traffic none 2> $O/r05_g_$tag.err | tail -1 > $O/r05_g_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r05_g_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()})
except Exception as e: print("$tag", "ERR", e)
EOF2
}
trio() { p=$1
  run ${p}_contig_one PBRT_AMD_DEVICE_LIB=$V/contig.so PBRT_AMD_SHADE_CLASSES=0
  run ${p}_inter_one PBRT_AMD_SHADE_CLASSES=0
  run ${p}_inter_parts A=1
}
WHAT=""; BARGS="--spp 16 --steps 3"; trio c3_16a; trio c3_16b
WHAT=""; BARGS="--steps 3"; trio c3_full
WHAT="--config c2"; BARGS="--steps 3"; trio c2_full
WHAT="--config c4"; BARGS="--steps 2"; trio c4_full
WHAT="--textured --leafmask"; BARGS="--spp 16 --steps 2"; trio texlm_16
WHAT="--subsurface"; BARGS="--spp 16 --steps 2"; trio sss_16
WHAT="--smokebox"; BARGS="--spp 16 --steps 2"; trio smoke_16
