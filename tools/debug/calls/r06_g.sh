#!/bin/bash
# round 6 call g: the Lambertian lobe 0 of constant materials in line (MatPack::diffuse0).  (1) GPU suite serial -x; (2) the default line; (3) C2 / C4 with their crops;
# (4) the phase profile of k_shade on C3 at 16 spp
cd /root/repo; O=gpurun_out; mkdir -p $O; T=${1:-r06_g}
timeout 1500 python -m pytest tests -x -q -m gpu > $O/${T}_pytest.txt 2>&1; tail -3 $O/${T}_pytest.txt
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/${T}_bench_c3.json 2> $O/${T}_bench_c3.err
python -c "
import json; d=json.loads(open('$O/${T}_bench_c3.json').read().strip().splitlines()[-1]); print('c3:', d['value'], d['kernel_ms_per_step'], d['cpu_baseline']['parity_crop']['pixels_within_tol'], 'secondary', d['secondary']['textured_leafmask']['value'], d['secondary']['textured_leafmask']['parity_crop']['pixels_within_tol'])"
for cfg in c2 c4; do
  timeout 900 python bench.py --config $cfg --steps 3 --warmup 1 --traffic none --secondary off --cpu-seconds 8 --cpu-port-seconds 0 > $O/${T}_bench_$cfg.json 2> $O/${T}_bench_$cfg.err
  python -c "
import json; d=json.loads(open('$O/${T}_bench_$cfg.json').read().strip().splitlines()[-1]); print('$cfg:', d['value'], d['kernel_ms_per_step'], d['cpu_baseline']['parity_crop'])"
done
PBRT_AMD_DEVICE_LIB=/root/repo/pbrt-v3-distributed_amd/lib/variants/shadeprof.so timeout 600 python bench.py --spp 16 --steps 1 --warmup 1 --traffic none --cpu-seconds 0 --secondary off > $O/${T}_prof_plain.json 2> $O/${T}_prof_plain.err
python -c "
import json; d=json.loads(open('$O/${T}_prof_plain.json').read().strip().splitlines()[-1]); print('plain under the profiler:', d['kernel_ms_per_step'])"
grep "shade-prof" $O/${T}_prof_plain.err | tail -23
