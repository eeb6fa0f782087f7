#!/bin/bash
# GPU checks that were still pending when round 1 ran out of GPU minutes (run this first next round):
#   1. the scenes pinned for the oracle only (thin-lens + Halton textures, PNG radiance map, the instancing stress scene in its default
#      flattened form) rendered on the device against their reference fixtures -- promote them to tests/test_gpu_parity.py when green;
#   1b. the differential fuzzer in device mode (device vs oracle on random scenes: tools/fuzz_vs_reference.py --device);
#   2. the experimental BVH8 traversal (tools/debug/bvh8.sh);
#   3. the textured shading kernel built for 2 waves per SIMD (-DPT_TEX_SHADE_WAVES=2): parity + the textured C3 probe (tools/debug/abtex.sh texw2
#      after building lib/variants/texw2.so with that flag).
#   4. the experimental two-level instancing traversal (k_trace / k_shade INST instances; PBRT_AMD_INSTANCING=1): device vs the reference
#      fixtures, which the two-level ORACLE reproduces bit for bit -- so the criterion to reach is exact equality up to the film sum.
cd /root/repo; mkdir -p gpurun_out
python - <<'PY' 2>&1 | tee gpurun_out/pending_scenes.txt
import os, sys, numpy as np
sys.path.insert(0, "tests")
import oracle_lib as ol, edge_scenes as es
pa = ol.pa
for name in es.TEX_ORACLE_ONLY + ["instances2"]:
    try:
        sc = pa.Scene(text=es.scene(name))
        ctx = pa.Context(sc); ctx.render()
        img = sc.film_image(ctx.film())
        ref = pa.read_pfm(os.path.join("tests", "golden", "edge_%s.pfm" % name))
        frac, relmse = ol.image_metrics(img, ref)
        print("%-14s frac %.4f relmse %.3e exact %.4f" % (name, frac, relmse, float(np.mean(np.abs(img - ref).max(-1) == 0))))
        ctx.close()
    except Exception as e:
        print("%-14s FAILED: %s" % (name, e))
PY
timeout 600 python tools/fuzz_vs_reference.py --device --n 60 --seed 7 --keep gpurun_out/fuzz_device 2>&1 | tail -15 | tee gpurun_out/pending_fuzz.txt
bash tools/debug/bvh8.sh

PBRT_AMD_INSTANCING=1 timeout 600 python - <<'PY' 2>&1 | tee gpurun_out/pending_instancing.txt
import os, sys, numpy as np
sys.path.insert(0, "tests")
import oracle_lib as ol, edge_scenes as es
pa = ol.pa
for name in es.INSTANCE_NAMES:
    try:
        sc = pa.Scene(text=es.scene(name))
        ctx = pa.Context(sc); ctx.render()
        img = sc.film_image(ctx.film())
        ref = pa.read_pfm(os.path.join("tests", "golden", "edge_%s.pfm" % name))
        frac, relmse = ol.image_metrics(img, ref)
        print("two-level %-12s frac %.4f relmse %.3e exact %.4f" % (name, frac, relmse, float(np.mean(np.abs(img - ref).max(-1) == 0))))
        ctx.close()
    except Exception as e:
        print("two-level %-12s FAILED: %s" % (name, e))
PY
