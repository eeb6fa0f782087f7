#!/usr/bin/env python3
"""Turn rocprofv3 output directories (gpurun_out/, scratch) into the small summaries kept under profiles/.

  python tools/profile_summary.py stats <dir> <out.csv>        copy the --kernel-trace --stats kernel table
  python tools/profile_summary.py pmc <dir> <out.json> [--traffic profiles/traffic_closest.json]
        per-kernel sums of every counter of a --pmc run; with --traffic also writes the HBM bytes per launch of the
        dominant kernel (k_trace<0,false,false>) from FETCH_SIZE: KB * 1024 * 2 (the gfx950 factor of
        guides/MI355X_MICROARCH.md, "HBM": FETCH_SIZE tallies 128-byte requests at 64 bytes)
"""
import csv, glob, json, os, shutil, sys, collections


def main():
    mode, d, out = sys.argv[1], sys.argv[2], sys.argv[3]
    if mode == "stats":
        f = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))
        if not f:
            raise SystemExit("no *kernel_stats.csv under " + d)
        shutil.copy(f[0], out)
        print("wrote", out)
        return
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k].add(r["Dispatch_Id"])
    res = {k: {"launches": len(launches[k]), **{c: v for c, v in cs.items()}} for k, cs in agg.items() if not k.startswith("__amd")}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out)
    if "--traffic" in sys.argv:
        tout = sys.argv[sys.argv.index("--traffic") + 1]
        key = [k for k in res if k.startswith("void k_trace<0, false, false>") or k.startswith("void k_trace<0, false>")]
        if key and "FETCH_SIZE" in res[key[0]]:
            kb = res[key[0]]["FETCH_SIZE"] / res[key[0]]["launches"]
            json.dump({"kernel": key[0], "launches": res[key[0]]["launches"], "FETCH_SIZE_KB_per_launch": kb,
                       "bytes_per_launch": kb * 1024 * 2,
                       "note": "rocprofv3 --pmc FETCH_SIZE (own pass); x2 = gfx950 correction of guides/MI355X_MICROARCH.md (HBM section); "
                               "includes Infinity-Cache hits; same bench command as the roofline (64 spp, one pass per frame)"},
                      open(tout, "w"), indent=1)
            print("wrote", tout)


if __name__ == "__main__":
    main()
