#!/bin/bash
# round 4, GPU call u: the smoke-box A/B of the walked shadow segments that call o missed (PBRT_AMD_TR_LEAN was not read for scenes without BSSRDF materials): 768-thread hot-node
# instance (shipped) against the sphere-capable one, 16 spp.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
for t in lean general; do
  if [ $t = general ]; then export PBRT_AMD_TR_LEAN=0; fi
  timeout 60 python bench.py --smokebox --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --cpu-port-seconds 0 --traffic none 2> $O/r04_u_$t.err | tail -1 > $O/r04_u_bench_smokebox_$t.json
  python -c "
import json; d=json.load(open('$O/r04_u_bench_smokebox_$t.json')); print('$t', d['value'], d['ms_per_step'], {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})"
done
