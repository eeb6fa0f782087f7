#!/bin/bash
# round 6 call m: compiler scheduling flags never measured before (LLVM AMDGPU: max-ilp / max-memory-clause strategies, schedule-metric-bias, wave priority, relaxed occupancy,
# divergent register indexing, spills to AGPRs) as lib/variants/cf_*.so against the shipped library: C3 plain, C3 textured + masked, C2, C4 (3 steps each, no CPU legs).
# Scheduling does not change arithmetic (-ffp-contract=off): the films must stay bit-identical (checked by the crop figures of a later call for whatever would ship).
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/pbrt-v3-distributed_amd/lib/variants; F=$O/r06_m_compiler_flags.txt; : > $F
one() { # name lib args...
  n=$1; lib=$2; shift 2
  if [ $lib = shipped ]; then unset PBRT_AMD_DEVICE_LIB; else export PBRT_AMD_DEVICE_LIB=$V/$lib.so; fi
  timeout 600 python bench.py "$@" --steps 3 --warmup 1 --traffic none --secondary off --cpu-seconds 0 --cpu-port-seconds 0 > $O/r06_m_$n.json 2> $O/r06_m_$n.err
  echo "$n rc $?: $(python - $O/r06_m_$n.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms', d['kernel_ms_per_step'])
except Exception as e: print('no line', e)
P
)" | tee -a $F
  unset PBRT_AMD_DEVICE_LIB
}
for lib in shipped cf_ilp cf_memclause cf_bias0 cf_wprio cf_relaxed cf_dividx cf_agpr1 cf_ilp_bias0 shipped; do one c3_$lib$([ -f $O/r06_m_c3_$lib.json ] && echo _again) $lib; done
for lib in shipped cf_ilp cf_memclause cf_bias0 cf_wprio cf_relaxed cf_dividx cf_agpr1; do one c3tex_$lib $lib --textured --leafmask; done
for lib in shipped cf_ilp cf_memclause cf_wprio cf_agpr1; do one c2_$lib $lib --config c2; one c4_$lib $lib --config c4; done
