#!/bin/bash
# round 2, last call: the general form of k_shade_vol on the volpath frame (PBRT_AMD_VOL_INLINE=1), f4 tests of the final library, a second fuzz campaign
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q -k "volpath or media or textured or instancing" 2>&1 | tail -2 | tee gpurun_out/r02t_pytest.txt
PBRT_AMD_VOL_INLINE=1 timeout 300 python bench.py --volpath --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --cpu-port-seconds 0 --traffic none 2> gpurun_out/r02t_volpath_general.err | tee gpurun_out/r02t_volpath_general.json | python -c "
import json,sys
d=json.load(sys.stdin); print('volpath general form 16spp', d['value'], d['kernel_ms_per_step'])" | tee gpurun_out/r02t_ab.txt
(
echo "== path"; timeout 300 python tools/fuzz_vs_reference.py --device --n 150 --seed 91 2>&1 | grep -v "^oracle:" | tail -3
echo "== volpath + media + subsurface"; timeout 300 python tools/fuzz_vs_reference.py --device --media --sss --n 150 --seed 92 2>&1 | grep -v "^oracle:" | tail -3
) > gpurun_out/r02t_fuzz.txt 2>&1; cat gpurun_out/r02t_fuzz.txt
