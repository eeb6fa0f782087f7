#!/bin/bash
# round 5, GPU call r: alpha masks pre-resolved per mesh (DevMaskFast: constants, dots and image maps over (u, v) answered without the node / program tables) -- parity of the masked
# paths, then the leaf-masked and the textured + leaf-masked C3 at 16 spp (call n's library: 169.2 Msamples/s textured + masked; call l: leaf-masked only 214.2).
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -n 4 -k "tex or alpha or fixture or vol or hot_nodes" 2>&1 | tail -2 | tee $O/r05_r_pytest.txt
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $WHAT $BARGS --warmup 1 --cpu-port-seconds 0 --cpu-seconds 0 --traffic none --secondary off 2> $O/r05_r_$tag.err | tail -1 > $O/r05_r_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r05_r_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()})
except Exception as e: print("$tag", "ERR", e)
EOF2
}
BARGS="--spp 16 --steps 2"
WHAT="--leafmask"; run lm A=1
WHAT="--textured --leafmask"; run texlm A=1
