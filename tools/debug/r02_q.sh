#!/bin/bash
# round 2: device fuzzing campaign of the final code (device vs oracle on random scenes; the oracle is pinned to the reference by the tool's default mode)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
(
echo "== path, two-level instancing (default)"; timeout 500 python tools/fuzz_vs_reference.py --device --n 220 --seed 81 2>&1 | grep -v "^oracle:" | tail -4
echo "== volpath + media"; timeout 400 python tools/fuzz_vs_reference.py --device --media --n 160 --seed 82 2>&1 | grep -v "^oracle:" | tail -4
echo "== volpath + media + subsurface"; timeout 400 python tools/fuzz_vs_reference.py --device --media --sss --n 160 --seed 83 2>&1 | grep -v "^oracle:" | tail -4
echo "== path + subsurface"; timeout 300 python tools/fuzz_vs_reference.py --device --sss --n 120 --seed 84 2>&1 | grep -v "^oracle:" | tail -4
echo "== volpath + media, instances flattened (PBRT_AMD_INSTANCING=0), general form forced (PBRT_AMD_VOL_INLINE=1)"; PBRT_AMD_INSTANCING=0 PBRT_AMD_VOL_INLINE=1 timeout 300 python tools/fuzz_vs_reference.py --device --media --n 100 --seed 85 2>&1 | grep -v "^oracle:" | tail -4
) > gpurun_out/r02q_fuzz.txt 2>&1
cat gpurun_out/r02q_fuzz.txt
