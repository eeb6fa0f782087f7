#!/bin/bash
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/pbrt-v3-distributed_amd/lib/variants
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $WHAT $BARGS --warmup 1 --cpu-port-seconds 0 --cpu-seconds 0 --traffic none --secondary off 2> $O/r05_o_$tag.err | tail -1 > $O/r05_o_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r05_o_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()})
except Exception as e: print("$tag", "ERR", e)
EOF2
}
BARGS="--spp 16 --steps 2"
WHAT="--textured"; run tex_default A=1; run tex_bump2 PBRT_AMD_DEVICE_LIB=$V/bump2.so; run tex_kd2 PBRT_AMD_DEVICE_LIB=$V/kd2.so
