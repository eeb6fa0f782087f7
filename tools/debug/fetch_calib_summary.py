"""Summarise the counter passes of tools/debug/fetch_calib.py: per dispatch of k_gather_probe / k_stream_read the counter value against the known byte count.
usage: fetch_calib_summary.py <dir with *counter_collection.csv> [GiB of the stream read]"""
import csv, glob, os, sys, collections
d = sys.argv[1]; gib = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
rows = collections.OrderedDict()
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_gather_probe" not in k and "k_stream_read" not in k: continue
        key = (int(r["Dispatch_Id"]), k.split("(")[0].replace("void ", ""), int(r.get("Grid_Size", 0) or 0))
        rows.setdefault(key, {})[r["Counter_Name"]] = rows.get(key, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for (disp, k, grid), c in sorted(rows.items()):
    if "gather" in k:
        loads = int(k.split("<")[1].split(">")[0])
        known = grid * 2048 * 64.0          # every lane fetches `iters` = 2048 records; a record is one 64-byte half of a 128-byte line (loads x 16 B of it are read)
        what = "random 64-B records (%d x 16 B read)" % loads
    else:
        known = gib * (1 << 30); what = "streaming read"
    s = "dispatch %4d %-22s %-36s known %8.3f GB" % (disp, k, what, known / 1e9)
    for n, v in sorted(c.items()):
        if n == "FETCH_SIZE": s += "  FETCH_SIZE %.3f GB (x%.3f to known)" % (v * 1024 / 1e9, known / (v * 1024) if v else float("nan"))
        else: s += "  %s %.4g (known/req = %.1f B)" % (n, v, known / v if v else float("nan"))
    print(s)
