#!/bin/bash
# round 5, GPU call x: LDS of the 768-thread traversal blocks re-divided between the per-lane stacks and the hot nodes -- 16 entries + 512 nodes (default), 14 + 608, 12 + 704 --
# C3 at 16 spp (twice), C4 at 32 spp; hit parity on the variants through the library switch.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/pbrt-v3-distributed_amd/lib/variants
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $WHAT $BARGS --warmup 1 --cpu-port-seconds 0 --cpu-seconds 0 --traffic none --secondary off 2> $O/r05_x_$tag.err | tail -1 > $O/r05_x_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r05_x_bench_$tag.json")); t = d.get("kernel_ms_per_step", {}); r = d["roofline"]
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()}, "hot share %.3f" % r.get("hot_share_of_node_visits", 0))
except Exception as e: print("$tag", "ERR", e)
EOF2
}
for v in hot608 hot704; do PBRT_AMD_DEVICE_LIB=$V/$v.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -n 4 -k "closest_hit or deep_stacks or hot_nodes" 2>&1 | tail -1; done
for rep in a b; do WHAT=""; BARGS="--spp 16 --steps 3"; run c3_16_${rep}_512 A=1; run c3_16_${rep}_608 PBRT_AMD_DEVICE_LIB=$V/hot608.so; run c3_16_${rep}_704 PBRT_AMD_DEVICE_LIB=$V/hot704.so; done
WHAT="--config c4"; BARGS="--spp 32 --steps 2"; run c4_32_512 A=1; run c4_32_608 PBRT_AMD_DEVICE_LIB=$V/hot608.so; run c4_32_704 PBRT_AMD_DEVICE_LIB=$V/hot704.so
