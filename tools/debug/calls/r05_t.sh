#!/bin/bash
# round 5, GPU call t: the key scan of the material sort with one block per key (was: one thread per key walking ~1 500 block histograms, 0.3 ms per bounce whatever the queue held) --
# C4 (maxdepth 30) and C3 at reduced spp, per-kernel stats of the C4 frame (rocprofv3 --kernel-trace --stats: where do the short bounces spend their time?).
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $WHAT $BARGS --warmup 1 --cpu-port-seconds 0 --cpu-seconds 0 --traffic none --secondary off 2> $O/r05_t_$tag.err | tail -1 > $O/r05_t_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r05_t_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()})
except Exception as e: print("$tag", "ERR", e)
EOF2
}
WHAT="--config c4"; BARGS="--spp 32 --steps 2"; run c4_32 A=1
WHAT=""; BARGS="--spp 16 --steps 3"; run c3_16 A=1
WHAT="--config c2"; BARGS="--spp 32 --steps 3"; run c2_32 A=1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/r05_t_prof -o c4 --output-format csv -- python $R/bench.py --config c4 --spp 32 --steps 2 --warmup 1 --cpu-seconds 0 --cpu-port-seconds 0 --traffic none --secondary off > /dev/null 2> $O/r05_t_prof.err)
python - <<'EOF2'
import csv, glob
for f in glob.glob("/root/repo/gpurun_out/r05_t_prof/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:22]: print("   %-70s calls %5s total %9.2f ms avg %8.1f us min %7.1f us" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
import collections
for f in glob.glob("/root/repo/gpurun_out/r05_t_prof/**/*kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    rows = [r for r in rows if "k_gather" not in r["Kernel_Name"] and "k_stream" not in r["Kernel_Name"]]
    # the last frame: from the last k_raygen on
    last = max(i for i, r in enumerate(rows) if "k_raygen" in r["Kernel_Name"])
    fr = rows[last:]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in fr); wall = int(fr[-1]["End_Timestamp"]) - int(fr[0]["Start_Timestamp"])
    print("last pass: %d launches, wall %.2f ms, kernels busy %.2f ms, gaps %.2f ms" % (len(fr), wall / 1e6, busy / 1e6, (wall - busy) / 1e6))
EOF2
rm -rf $O/r05_t_prof
