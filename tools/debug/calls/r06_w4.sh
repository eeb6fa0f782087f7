#!/bin/bash
# round 6 call w4 (measurement only): is it the REQUEST or the LINE that costs?  fifth_same.so (PT_FIFTH_LOAD=4): the cold lanes load the last word of their own node a second time (same 128-byte line, no new miss)
# against fifth_glb.so (the first word of the next node: a new line for every second node).
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
B="python bench.py --steps 6 --warmup 2 --cpu-seconds 0 --cpu-port-seconds 0 --traffic none --secondary off"
run() { name=$1; shift; env "$@" timeout 600 $B > $O/r06_w4_$name.json 2> $O/r06_w4_$name.err; python - $O/r06_w4_$name.json $name <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print("%-18s %8.1f Msamples/s  closest %7.2f  anyhit %6.2f  shade %7.2f ms per frame" % (sys.argv[2], d['value'], k['closest'], k['anyhit'], k['shade']))
P
}
V=$R/pbrt-v3-distributed_amd/lib/variants
run shipped_a A=1
run fifth_same PBRT_AMD_DEVICE_LIB=$V/fifth_same.so
run fifth_glb PBRT_AMD_DEVICE_LIB=$V/fifth_glb.so
run shipped_b A=1
