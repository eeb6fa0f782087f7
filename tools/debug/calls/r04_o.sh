#!/bin/bash
# round 4, GPU call o: subsurface scenes -- (a) only the vertices on BSSRDF materials through k_shade_vol, the rest through k_shade (PBRT_AMD_SSS_ROUTE, default on),
# (b) the probe / shadow walks of all-triangle scenes on the 768-thread hot-node instance (PBRT_AMD_TR_LEAN, default on).  Subsurface / vol parity tests, then
# bench.py --subsurface at 16 spp in the four combinations, per-kernel times of the shipped form (rocprofv3 --kernel-trace --stats), 64 spp with the pbrt_ref crop, and the smoke box (walked shadow rays).
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "subsurface or bssrdf or walked or sss or vol or only_the_vertices" 2>&1 | tail -2
run() { tag=$1; shift; env "$@" timeout 400 python bench.py $WHAT $BARGS --steps 2 --warmup 1 --cpu-port-seconds 0 --traffic none 2> $O/r04_o_$tag.err | tail -1 > $O/r04_o_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r04_o_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()}, (d.get("cpu_baseline") or {}).get("parity_crop", {}).get("pixels_within_tol"))
except Exception as e: print("$tag", "ERR", e)
EOF2
}
WHAT=--subsurface; BARGS="--spp 16 --cpu-seconds 0"
run sss_shipped_16spp A=1
run sss_all_vol_16spp PBRT_AMD_SSS_ROUTE=0
run sss_tr_general_16spp PBRT_AMD_TR_LEAN=0
run sss_round3_form_16spp PBRT_AMD_SSS_ROUTE=0 PBRT_AMD_TR_LEAN=0
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/r04_o_prof -o sss --output-format csv -- python $R/bench.py --subsurface --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --cpu-port-seconds 0 --traffic none > $O/r04_o_bench_sss_under_rocprof.json 2> $O/r04_o_prof.err)
head -14 $O/r04_o_prof/sss_kernel_stats.csv | cut -c1-170
BARGS="--cpu-seconds 10"
run sss_shipped A=1
WHAT=--smokebox; BARGS="--spp 16 --cpu-seconds 0"
run smokebox_shipped_16spp A=1
run smokebox_tr_general_16spp PBRT_AMD_TR_LEAN=0
