#!/bin/bash
# round 2, first GPU call: everything round 1 left unrun on hardware (oracle-only scenes, BVH8 kernels, INST kernels, device fuzz)
cd /root/repo; mkdir -p gpurun_out
python - <<'PY' 2>&1 | tee gpurun_out/r02a_scenes.txt
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
PBRT_AMD_BVH8=1 timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/r02a_bvh8_pytest.txt
run() { timeout 200 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 2>gpurun_out/r02a_$1.err | tee gpurun_out/r02a_$1.json | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('$1', d['value'], d['kernel_ms_per_step'], 'nodes/ray', round(r['nodes_per_ray'],2), 'tris/ray', round(r['tris_per_ray'],2))"; }
run bvh4 | tee gpurun_out/r02a_ab.txt
PBRT_AMD_BVH8=1 run bvh8 | tee -a gpurun_out/r02a_ab.txt
PBRT_AMD_INSTANCING=1 timeout 300 python - <<'PY' 2>&1 | tee gpurun_out/r02a_instancing.txt
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
timeout 240 python tools/fuzz_vs_reference.py --device --n 30 --seed 7 --keep gpurun_out/fuzz_device 2>&1 | tail -15 | tee gpurun_out/r02a_fuzz.txt
