#!/bin/bash
# round 5, GPU call k: the masked traversal instances on the MID shape (two 512-thread blocks per CU, 128-VGPR cap, hot nodes in LDS; mid0 = round 2's shape, mid640 / mid768 =
# 5 / 6 waves per SIMD with more spills) and the texture program loop with its arithmetic nodes in line -- textured + leaf-masked, leaf-masked only, textured only C3 at 16 spp;
# parity tests of the masked / textured paths on the hardware.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/pbrt-v3-distributed_amd/lib/variants
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -n 4 -k "tex or alpha or material or fixture or closest_hit" 2>&1 | tail -2 | tee $O/r05_k_pytest.txt
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $WHAT $BARGS --warmup 1 --cpu-port-seconds 0 --cpu-seconds 0 --traffic none 2> $O/r05_k_$tag.err | tail -1 > $O/r05_k_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r05_k_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()})
except Exception as e: print("$tag", "ERR", e)
EOF2
}
BARGS="--spp 16 --steps 2"
WHAT="--leafmask"; run lm_default A=1; for v in mid0 mid640 mid768; do run lm_$v PBRT_AMD_DEVICE_LIB=$V/$v.so; done
WHAT="--textured --leafmask"; run texlm_default A=1; run texlm_mid0 PBRT_AMD_DEVICE_LIB=$V/mid0.so
WHAT="--textured"; run tex_default A=1
