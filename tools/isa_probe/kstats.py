"""Per-kernel resource figures from a device assembly listing (hipcc --cuda-device-only -S): VGPRs, SGPR / VGPR spills, scratch bytes, code bytes are read from the
.amdhsa metadata at the end of the listing.  usage: kstats.py dev.s [substring ...]"""
import re, sys, subprocess
txt = open(sys.argv[1]).read()
pats = sys.argv[2:]
m = re.search(r"amdhsa.kernels:(.*?)amdhsa.target", txt, re.S)
kern = re.split(r"\n  - ", m.group(1))
def dem(n):
    try: return subprocess.run(["/usr/bin/c++filt", n], stdout=subprocess.PIPE, text=True).stdout.strip()
    except Exception: return n
for k in kern:
    g = lambda key: (re.search(r"\." + key + r":\s*(\S+)", k) or [None, "?"])[1]
    name = dem(g("name"))
    if pats and not any(p in name for p in pats): continue
    print("%-110s vgpr %4s sgpr %4s sspill %4s vspill %4s scratch %6s lds %6s" % (name.replace("void ", "")[:110], g("vgpr_count"), g("sgpr_count"), g("sgpr_spill_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
