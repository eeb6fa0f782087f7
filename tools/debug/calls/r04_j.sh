#!/bin/bash
# round 4, GPU call j: the tail of the BSSRDF probe walk in one launch (k_sss_probe_tail: queue below PBRT_AMD_SSS_TAIL rays, default 65536) against rounds to the
# end (PBRT_AMD_SSS_TAIL=0) -- the subsurface tests, then bench.py --subsurface at 16 spp (both, and thresholds 4096 / 1 M) and at 64 spp with the pbrt_ref crop.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "subsurface or bssrdf or walked or sss or vol" 2>&1 | tail -2
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --subsurface $BARGS --steps 2 --warmup 1 --cpu-port-seconds 0 --traffic none 2> $O/r04_j_$tag.err | tail -1 > $O/r04_j_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r04_j_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()}, (d.get("cpu_baseline") or {}).get("parity_crop", {}).get("pixels_within_tol"))
except Exception as e: print("$tag", "ERR", e)
EOF2
}
BARGS="--spp 16 --cpu-seconds 0"
run tail65536_16spp A=1
run rounds_16spp PBRT_AMD_SSS_TAIL=0
run tail4096_16spp PBRT_AMD_SSS_TAIL=4096
run tail1M_16spp PBRT_AMD_SSS_TAIL=1048576
BARGS="--cpu-seconds 10"
run tail65536 A=1
