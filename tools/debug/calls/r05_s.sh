#!/bin/bash
# round 5, GPU call s: per-lane lobe records fetched whole with wide loads before the lobe switch (PT_LOBE_WIDE; lobe0 = lib/variants/lobe0.so = field by field where used) --
# textured, textured + leaf-masked, subsurface and smoke-box C3 at 16 spp; BxDF / material / texture parity on the hardware.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/pbrt-v3-distributed_amd/lib/variants
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -n 4 -k "tex or bxdf or material or fixture or vol or sss" 2>&1 | tail -2 | tee $O/r05_s_pytest.txt
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $WHAT $BARGS --warmup 1 --cpu-port-seconds 0 --cpu-seconds 0 --traffic none --secondary off 2> $O/r05_s_$tag.err | tail -1 > $O/r05_s_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r05_s_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()})
except Exception as e: print("$tag", "ERR", e)
EOF2
}
BARGS="--spp 16 --steps 2"
for w in "--textured" "--textured --leafmask" "--subsurface" "--smokebox"; do WHAT="$w"; n=$(echo $w | tr -d ' -'); run ${n}_lobe0 PBRT_AMD_DEVICE_LIB=$V/lobe0.so; run ${n}_wide A=1; done
