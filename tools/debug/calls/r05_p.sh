#!/bin/bash
# round 5, GPU call p: the library's own topology over the reference's leaves (pt_treebuild.h; PBRT_AMD_TREE=reference = the tree as handed over) -- hit-for-hit parity in every
# traversal mode, then C3 (16 spp twice, full size), C2, C4, the textured + leaf-masked C3; nodes / triangles per ray from the counting pass of bench.py.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -n 4 -k "closest_hit or fixture or baseline_configs or deep_stacks" 2>&1 | tail -2 | tee $O/r05_p_pytest.txt
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $WHAT $BARGS --warmup 1 --cpu-port-seconds 0 --cpu-seconds 0 --traffic none --secondary off 2> $O/r05_p_$tag.err | tail -1 > $O/r05_p_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r05_p_bench_$tag.json")); t = d.get("kernel_ms_per_step", {}); r = d["roofline"]
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()}, "nodes/ray %.2f tris/ray %.2f hot %.3f upload %.1f s" % (r.get("nodes_per_ray", 0), r.get("tris_per_ray", 0), r.get("hot_share_of_node_visits", 0), d["setup_s"]["upload_and_bvh4"]))
except Exception as e: print("$tag", "ERR", e)
EOF2
}
pair() { run $1_ref PBRT_AMD_TREE=reference; run $1_own A=1; }
WHAT=""; BARGS="--spp 16 --steps 3"; pair c3_16a; pair c3_16b
BARGS="--steps 3"; pair c3_full
WHAT="--config c2"; pair c2_full
WHAT="--config c4"; BARGS="--steps 2"; pair c4_full
WHAT="--textured --leafmask"; BARGS="--spp 16 --steps 2"; pair texlm_16
