#!/bin/bash
# round 5, GPU call e: run-to-run spread of k_shade and what each material class costs with and without its instance -- C3 at 16 spp under rocprofv3 --kernel-trace, three modes
# alternating twice on ONE box: class instances (default), the same parts shaded by the generic instance (PBRT_AMD_SHADE_CLASSES=generic), one generic launch (=0).
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
for rep in 1 2; do for mode in cls generic off; do
  case $mode in cls) export PBRT_AMD_SHADE_CLASSES=1;; generic) export PBRT_AMD_SHADE_CLASSES=generic;; off) export PBRT_AMD_SHADE_CLASSES=0;; esac
  tag=${mode}_$rep
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/r05_e_prof_$tag -o c3 --output-format csv -- python $R/bench.py --spp 16 --steps 3 --warmup 1 --cpu-seconds 0 --cpu-port-seconds 0 --traffic none > $O/r05_e_bench_$tag.json 2> $O/r05_e_$tag.err)
  python - <<EOF2
import json, csv, glob
d = json.load(open("$O/r05_e_bench_$tag.json")); print("$tag", d["value"], d["ms_per_step"], d["kernel_ms_per_step"]["shade"])
for f in glob.glob("$O/r05_e_prof_$tag/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_shade" in r["Kernel_Name"]]
    per = {}
    for i, r in enumerate(rows): per.setdefault(r["Kernel_Name"].split("(")[0], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in per.items(): print("   %-50s launches %3d  total %9.1f us   first six: %s" % (k, len(v), sum(v), " ".join("%.0f" % x for x in v[:6])))
EOF2
  rm -rf $O/r05_e_prof_$tag
done; done
