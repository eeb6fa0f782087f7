#!/bin/bash
# round 4, GPU call n: the scheduling thresholds of k_trace re-swept after the interior step and the triangle phase got shorter (207 / 205 instructions per round instead of
# 250 / 264): leaf phase at 16 / 20 / 24 (shipped) / 32 waiting lanes, at most 4 / 8 (shipped) / 12 node steps between two leaf phases, refill at 12 / 16 (shipped) idle lanes.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
run() { tag=$1; shift; env "$@" timeout 100 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none 2> $O/r04_n_$tag.err | tail -1 > $O/r04_n_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r04_n_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()})
except Exception as e: print("$tag", "ERR", e)
EOF2
}
V=$R/pbrt-v3-distributed_amd/lib/variants
run shipped A=1
for v in L16 L20 L32 N4 N12 R12; do run $v PBRT_AMD_DEVICE_LIB=$V/$v.so; done
run shipped_again A=1
