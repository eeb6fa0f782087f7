#!/usr/bin/env python3
"""Compile tools/isa_probe/*.hip for gfx950 (device only, -S) and print per-kernel instruction counts by class."""
import os, re, subprocess, sys, tempfile
here = os.path.dirname(os.path.abspath(__file__))
for src in sorted(f for f in os.listdir(here) if f.endswith(".hip")):
    out = os.path.join(tempfile.mkdtemp(), "p.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S",
                           os.path.join(here, src), "-o", out], stderr=subprocess.DEVNULL)
    s = open(out).read()
    print("== %s" % src)
    for m in re.finditer(r'^(\w+):\s*;\s*@\1\n(.*?)^\.Lfunc_end', s, re.S | re.M):
        body = [l.strip().split()[0] for l in m.group(2).split("\n") if l.strip() and not l.strip().startswith((";", ".")) and not l.split(";")[0].strip().endswith(":")]
        cls = {"valu": 0, "salu": 0, "vmem_load": 0, "vmem_store": 0, "lds": 0, "other": 0}
        for op in body:
            if op.startswith(("global_load", "flat_load", "buffer_load", "scratch_load")): cls["vmem_load"] += 1
            elif op.startswith(("global_store", "flat_store", "buffer_store", "scratch_store")): cls["vmem_store"] += 1
            elif op.startswith("ds_"): cls["lds"] += 1
            elif op.startswith("v_"): cls["valu"] += 1
            elif op.startswith("s_"): cls["salu"] += 1
            else: cls["other"] += 1
        vg = re.search(r'\.amdhsa_next_free_vgpr (\d+)', s[m.end():m.end() + 4000])
        print("  %-12s %s  vgpr %s" % (m.group(1), cls, vg.group(1) if vg else "?"))
