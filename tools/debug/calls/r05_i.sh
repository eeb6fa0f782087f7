#!/bin/bash
# round 5, GPU call i: the textured + leaf-masked C3 at 16 spp -- where does k_shade<..., TEX> lose its time?  Variants: mix1 = PT_MIX_MAX_DEPTH=1 (no mix recursion frames: an
# upper bound for what a no-mix instance buys; the scene has no mix material), texw3 = 3 waves per SIMD (168 VGPRs), texuni = PT_TEX_UNIFORM=1 (wave-uniform texture / material
# tables through scalar loads), mix1uni = both; + rocprofv3 kernel stats of the default build.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/pbrt-v3-distributed_amd/lib/variants
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $WHAT $BARGS --warmup 1 --cpu-port-seconds 0 --cpu-seconds 0 --traffic none 2> $O/r05_i_$tag.err | tail -1 > $O/r05_i_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r05_i_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()})
except Exception as e: print("$tag", "ERR", e)
EOF2
}
WHAT="--textured --leafmask"; BARGS="--spp 16 --steps 2"
run texlm_default A=1
for v in mix1 texw3 texuni mix1uni; do run texlm_$v PBRT_AMD_DEVICE_LIB=$V/$v.so; done
WHAT="--textured"; run tex_default A=1; run tex_mix1uni PBRT_AMD_DEVICE_LIB=$V/mix1uni.so
WHAT="--leafmask"; run lm_default A=1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/r05_i_prof -o t --output-format csv -- python $R/bench.py --textured --leafmask --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --cpu-port-seconds 0 --traffic none > /dev/null 2> $O/r05_i_prof.err)
head -12 $O/r05_i_prof/t_kernel_stats.csv | cut -c1-200
rm -f $O/r05_i_prof/t_kernel_trace.csv
