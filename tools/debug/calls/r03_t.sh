#!/bin/bash
# round 3, GPU call t: the alpha phases' trigger.  Call s: threshold 8 / 16 / 32 lanes = 111 / 138 / 150 Msamples/s on the 16-spp masked-leaves frame
# (inline evaluation 67) -- waiting is cheap, the mask interpreter is not.  Here: thresholds 32 (shipped) / 48 / 64 and a more patient second rule
# (phase when the waiting lanes outnumber PT_ALPHA_GO_MUL x the lanes that can go on; 64 = second rule only).
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
run() { tag=$1; shift; env "$@" 2> $O/r03_t_$tag.err | tail -1 > $O/r03_t_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r03_t_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 1) for k, v in t.items()} if isinstance(t, dict) else t)
except Exception as e: print("$tag", "ERR", e)
EOF2
}
V=$R/pbrt-v3-distributed_amd/lib/variants
P="--leafmask --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none"
run lm16_a32 A=1 timeout 300 python bench.py $P
for v in a48 a64 a32m2 a48m2 a64m4; do run lm16_$v PBRT_AMD_DEVICE_LIB=$V/$v.so timeout 300 python bench.py $P; done
