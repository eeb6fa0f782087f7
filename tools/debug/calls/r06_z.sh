#!/bin/bash
# round 6 call z: the committed state -- GPU suite (serial, -x, as the driver runs it), smoke, the default line (saved as the traffic file's source), every other BASELINE
# config at its quoted size with its pbrt_ref crop, the variants of C3, rocprofv3 kernel stats of the default command, the 2-rank job 10 x, a device-vs-oracle fuzz campaign.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R; T=${1:-r06_z}
timeout 1500 python -m pytest tests -x -q -m gpu --durations=10 > $O/${T}_pytest.txt 2>&1; tail -2 $O/${T}_pytest.txt
( time timeout 200 python __graft_entry__.py smoke ) 2>&1 | grep -v "^$\|user\|sys" | tee -a $O/${T}_pytest.txt
timeout 1500 python bench.py --steps 20 --warmup 5 --save-traffic > $O/${T}_bench_c3.json 2> $O/${T}_bench_c3.err; cp profiles/traffic_closest.json $O/${T}_traffic_closest.json
line() { python - "$1" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}; pc=(d.get('cpu_baseline') or {}).get('parity_crop') or {}
print(sys.argv[1].split('/')[-1], d['value'], 'Msamples/s', d['ms_per_step'], 'ms', d['kernel_ms_per_step'], 'frac', r.get('frac'), 'closest launch ms', round(r.get('avg_launch_ms') or 0, 2), 'crop', pc.get('pixels_within_tol'), pc.get('pixels'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'secondary', ((d.get('secondary') or {}).get('textured_leafmask') or {}).get('value'))
P
}
line $O/${T}_bench_c3.json
for spec in "c2:--config c2" "c4:--config c4" "c5:--config c5 --steps 1 --warmup 1" "c3_subsurface:--subsurface" "c3_smokebox:--smokebox" "c3_textured_leafmask:--textured --leafmask"; do
  name=${spec%%:*}; args=${spec#*:}
  case "$args" in *--steps*) S="";; *) S="--steps 3 --warmup 1";; esac
  timeout 1200 python bench.py $args $S --traffic none --secondary off --cpu-seconds 8 --cpu-port-seconds 0 > $O/${T}_bench_$name.json 2> $O/${T}_bench_$name.err
  line $O/${T}_bench_$name.json
done
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/${T}_prof -o c3 --output-format csv -- python $R/bench.py --cpu-seconds 0 --traffic none --secondary off > $O/${T}_bench_c3_under_rocprof.json 2> $O/${T}_prof.err)
python tools/profile_summary.py stats $O/${T}_prof $O/${T}_kernel_stats.csv; rm -rf $O/${T}_prof
LIMIT=200 tools/debug/n2_loop.sh /root/repo 10 ${T} > $O/${T}_n2_loop.txt 2>&1; tail -2 $O/${T}_n2_loop.txt
F=$O/${T}_device_fuzz_campaign.txt; : > $F
fz() { echo "== $*  ${ARGS}" | tee -a $F; env "$@" timeout 500 python tools/fuzz_vs_reference.py --device ${ARGS} 2>&1 | tail -2 | tee -a $F; }
ARGS="--n 220 --seed 901" fz PBRT_AMD_INSTANCING=0
ARGS="--n 90 --seed 902" fz A=1
ARGS="--media --sss --n 110 --seed 903" fz A=1
ARGS="--media --n 80 --seed 904" fz PBRT_AMD_INSTANCING=0
