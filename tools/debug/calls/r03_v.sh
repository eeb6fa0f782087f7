#!/bin/bash
# round 3, GPU call v (the round's last GPU seconds): the leaf-phase thresholds again.  Round 2 swept them (leaf phase at 16 / 32 / 40 waiting lanes, 4 / 16 / 32
# node steps between two leaf phases) BEFORE the parked leaves; since then a lane with a parked leaf goes on with node steps, so waiting for a fuller leaf
# phase costs less -- as the alpha phases just showed (threshold 8 / 16 / 32 lanes = 111 / 138 / 150).  16-spp C3 probe frame, shipped (24 lanes, 8 steps) vs variants.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
run() { tag=$1; shift; env "$@" timeout 100 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none 2> $O/r03_v_$tag.err | tail -1 > $O/r03_v_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r03_v_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()})
except Exception as e: print("$tag", "ERR", e)
EOF2
}
V=$R/pbrt-v3-distributed_amd/lib/variants
run L24N8_shipped A=1
for v in L32 L40 L48 L32N16 L40N16 L48N16; do run $v PBRT_AMD_DEVICE_LIB=$V/$v.so; done
