#!/bin/bash
# round 5, GPU call m: (1) PBRT_AMD_OVERLAP=1 (direct-lighting traversals of a bounce on a second stream, overlapping the next bounce's path-extension traversal) re-measured now that
# the queues are balanced -- C3, C2, C4 at full size; (2) the San-Miguel-like variants at full size on the current library.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $WHAT $BARGS --warmup 1 --cpu-port-seconds 0 --cpu-seconds 0 --traffic none --secondary off 2> $O/r05_m_$tag.err | tail -1 > $O/r05_m_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r05_m_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()})
except Exception as e: print("$tag", "ERR", e)
EOF2
}
WHAT=""; BARGS="--steps 3"; run c3_full A=1; run c3_full_overlap PBRT_AMD_OVERLAP=1
WHAT="--config c2"; run c2_full A=1; run c2_full_overlap PBRT_AMD_OVERLAP=1
WHAT="--config c4"; BARGS="--steps 2"; run c4_full A=1; run c4_full_overlap PBRT_AMD_OVERLAP=1
BARGS="--steps 2"
WHAT="--textured --leafmask"; run texlm_full A=1; run texlm_full_overlap PBRT_AMD_OVERLAP=1
WHAT="--subsurface"; run sss_full A=1
WHAT="--smokebox"; run smoke_full A=1
