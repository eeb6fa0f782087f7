#!/bin/bash
# round 6 call t: k_shade reloads the hit triangle (vertices, normals / uvs, flags) from ONE 128-byte line (DevScene::tri_rec) instead of three arrays
# (lib/variants/notrirec.so = the three arrays): GPU suite, then C3 / textured C3 / C2 / C4 A/B/A/B
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R; T=r06_t
timeout 1500 python -m pytest tests -x -q -m gpu --durations=5 > $O/${T}_pytest.txt 2>&1; tail -2 $O/${T}_pytest.txt
line() { python - "$1" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); pc=(d.get('cpu_baseline') or {}).get('parity_crop') or {}
    print(sys.argv[1].split('/')[-1], d['value'], 'Msamples/s', d['ms_per_step'], 'ms', d['kernel_ms_per_step'], 'crop', pc.get('pixels_within_tol'), pc.get('pixels'))
except Exception as e: print(sys.argv[1], 'no line', e)
P
}
B="--steps 3 --warmup 1 --traffic none --secondary off --cpu-port-seconds 0"
for spec in "c3:" "tex:--textured --leafmask" "c2:--config c2" "c4:--config c4"; do
  name=${spec%%:*}; args=${spec#*:}
  for lib in new old new2 old2; do
    case $lib in old*) export PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/notrirec.so; CPU=0;; new) unset PBRT_AMD_DEVICE_LIB; CPU=8;; *) unset PBRT_AMD_DEVICE_LIB; CPU=0;; esac
    timeout 900 python bench.py $args $B --cpu-seconds $CPU > $O/${T}_bench_${name}_$lib.json 2> $O/${T}_bench_${name}_$lib.err; line $O/${T}_bench_${name}_$lib.json
  done
done
