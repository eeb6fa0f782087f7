#!/bin/bash
# round 6 call w2 (measurement only): what a fifth 16-byte request per interior step costs the traversal kernels on C3 (lib/variants/fifth.so = -DPT_FIFTH_LOAD=1: the first
# word of the next node loaded and waited for with the node words, unused) -- the price of an 80-byte node, to set against the ~10 % of the step's issue cycles f16 planes
# through v_fma_mix_f32 would save (profiles/r06_w_f16_node_probe.txt).  Shipped library first and last (box drift), the variant in between.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
B="python bench.py --steps 6 --warmup 2 --cpu-seconds 0 --cpu-port-seconds 0 --traffic none --secondary off"
run() { name=$1; shift; env "$@" timeout 600 $B > $O/r06_w2_$name.json 2> $O/r06_w2_$name.err; python - $O/r06_w2_$name.json $name <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print("%-18s %8.1f Msamples/s  closest %7.2f  anyhit %6.2f  shade %7.2f ms per frame  nodes/ray %.2f" % (sys.argv[2], d['value'], k['closest'], k['anyhit'], k['shade'], d['roofline']['nodes_per_ray']))
P
}
run shipped_a A=1
run fifth PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/fifth.so
run shipped_b A=1
run fifth_b PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/fifth.so
