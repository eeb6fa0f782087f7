#!/bin/bash
# round 6 call w3 (measurement only): which pipe charges for the fifth request of call w2 -- the extra ds_read_b128 of the hot lanes alone (fifth_lds.so, PT_FIFTH_LOAD=2)
# or the extra global load of the cold lanes alone (fifth_glb.so, PT_FIFTH_LOAD=3)?
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
B="python bench.py --steps 6 --warmup 2 --cpu-seconds 0 --cpu-port-seconds 0 --traffic none --secondary off"
run() { name=$1; shift; env "$@" timeout 600 $B > $O/r06_w3_$name.json 2> $O/r06_w3_$name.err; python - $O/r06_w3_$name.json $name <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print("%-18s %8.1f Msamples/s  closest %7.2f  anyhit %6.2f  shade %7.2f ms per frame" % (sys.argv[2], d['value'], k['closest'], k['anyhit'], k['shade']))
P
}
V=$R/pbrt-v3-distributed_amd/lib/variants
run shipped_a A=1
run fifth_lds PBRT_AMD_DEVICE_LIB=$V/fifth_lds.so
run fifth_glb PBRT_AMD_DEVICE_LIB=$V/fifth_glb.so
run shipped_b A=1
