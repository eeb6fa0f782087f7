#!/bin/bash
# C3 frame with pbrt's default sampler (halton) instead of sobol: performance of the Halton path
cd /root/repo; mkdir -p gpurun_out /tmp/h
python tools/gen_scenes.py sanmiguel --tris 10000000 --res 1920 1080 --spp 16 --out /tmp/h/sm.pbrt > /dev/null
sed -i 's/Sampler "sobol"/Sampler "halton"/' /tmp/h/sm.pbrt
grep -n "Sampler" /tmp/h/sm.pbrt
timeout 300 python bench.py --scene /tmp/h/sm.pbrt --steps 2 --warmup 1 --cpu-seconds 0 2>gpurun_out/halton.err | python -c "
import json,sys
d=json.load(sys.stdin); print('halton', d['value'], d['mrays_per_s'], d['kernel_ms_per_step'])"
