#!/bin/bash
# round 5, GPU call d: why did the class launches of C3 lose between call b (shade 30.9 ms at 16 spp) and call c (40.7)?  Per-kernel durations (rocprofv3 --kernel-trace --stats) of the
# current library and of call b's library (lib/variants/bref.so, built from commit 5e3ce6a with -DPT_SHADE_CLS_WAVES=3) on the same box.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/pbrt-v3-distributed_amd/lib/variants
for tag in now bref; do
  if [ $tag = bref ]; then export PBRT_AMD_DEVICE_LIB=$V/bref.so; fi
  timeout 300 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-port-seconds 0 --cpu-seconds 0 --traffic none 2> $O/r05_d_$tag.err | tail -1 > $O/r05_d_bench_$tag.json
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/r05_d_prof_$tag -o c3 --output-format csv -- python $R/bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --cpu-port-seconds 0 --traffic none > $O/r05_d_bench_${tag}_under_rocprof.json 2> $O/r05_d_prof_$tag.err)
  python - <<EOF2
import json, csv, glob
d = json.load(open("$O/r05_d_bench_$tag.json")); print("$tag", d["value"], d["ms_per_step"], d["kernel_ms_per_step"])
for f in glob.glob("$O/r05_d_prof_$tag/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]: print("   %-80s calls %5s total %10.3f ms avg %9.1f us" % (r["Name"][:80], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
EOF2
  find $O/r05_d_prof_$tag -name "*kernel_trace.csv" -size +3M -delete
done
