#!/usr/bin/env python3
"""Memory-side read traffic per kernel of a short bench run, from one rocprofv3 --pmc pass (GPU box): bytes = 32 n32 + 64 n64 + 128 n128 of the L2's read requests
(TCC_EA0_RDREQ_*; calibrated in round 5, profiles/r05_a_*).  Usage: kernel_traffic.py <tag> [bench args ...]; PBRT_AMD_DEVICE_LIB selects a variant."""
import collections, csv, glob, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag, extra = sys.argv[1], sys.argv[2:]
# (call v asked for the two WRREQ counters in the same pass: six TCC counters never finished -- 900 s, no output; the four below are bench.py's own pass, 18 s)
names = ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"]
d = tempfile.mkdtemp(prefix="pbrt_amd_kt_", dir="/tmp")
cmd = ["rocprofv3", "--pmc"] + names + ["-d", d, "-o", "c", "--output-format", "csv", "--", sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0",
       "--cpu-seconds", "0", "--cpu-port-seconds", "0", "--traffic", "none", "--secondary", "off", "--pmc-child"] + extra
r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=int(os.environ.get("KT_TIMEOUT_S", "300")))
if r.returncode != 0:
    sys.exit("PMC pass failed: " + r.stdout[-500:])
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if row.get("Counter_Name") in names:
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            cnt[k][row["Counter_Name"]] += float(row["Counter_Value"]); disp[k].add(row["Dispatch_Id"])
shutil.rmtree(d, ignore_errors=True)
print("# %s: %s" % (tag, " ".join(extra)))
print("%-58s %8s %14s %14s" % ("kernel", "launches", "read GB/frame", "read Mreq/frame"))
for k in sorted(cnt, key=lambda k: -cnt[k]["TCC_EA0_RDREQ_sum"])[:10]:
    c = cnt[k]
    rd = 32.0 * c["TCC_EA0_RDREQ_32B_sum"] + 64.0 * c["TCC_EA0_RDREQ_64B_sum"] + 128.0 * c["TCC_EA0_RDREQ_128B_sum"]
    print("%-58s %8d %14.2f %14.1f" % (k[:58], len(disp[k]), rd * 1e-9, c["TCC_EA0_RDREQ_sum"] * 1e-6))
