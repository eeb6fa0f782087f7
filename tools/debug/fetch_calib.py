"""FETCH_SIZE calibration: run the two probe kernels with a KNOWN byte count per launch -- k_gather_probe<4> (dependent random
64-byte record fetches, 4 x 16 B per lane: the access pattern of one BVH4Q node step) and k_stream_read (wide coalesced streaming read) -- over a buffer far
beyond the 256 MiB Infinity Cache.  Run it under `rocprofv3 --pmc FETCH_SIZE` (and a second pass with TCC_EA0_RDREQ TCC_EA0_RDREQ_32B); tools/debug/fetch_calib_summary.py
divides the counter values by the known bytes.  usage: fetch_calib.py [GiB]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pa = importlib.import_module("pbrt-v3-distributed_amd")
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
sc = pa.Scene(text=open(os.path.join(ROOT, "scenes", "cornell.pbrt")).read())
ctx = pa.Context(sc, device=0)
for loads in (4, 1):
    r = ctx.gather_rate(int(gib * (1 << 30)), loads)
    print("gather_probe<%d> over %.1f GiB: %.2f G lane requests/s = %.2f TB/s of 16-byte lane loads" % (loads, gib, r, r * 16e-3))
print("stream_read over %.1f GiB: %.1f GB/s" % (gib, ctx.stream_read_gbps(int(gib * (1 << 30)))))
# in-cache partner: the same gather over 64 MiB (inside L2 + Infinity Cache): FETCH_SIZE counts what leaves the L2
r = ctx.gather_rate(64 << 20, 4)
print("gather_probe<4> over 64 MiB: %.2f G lane requests/s" % r)
ctx.close()
