#!/bin/bash
# round 6 call x: HEAD after call w (bench.py's roofline_shade note names the memory-side figure; nothing else in the product changed since call zzz) --
# the GPU suite serially with -x as the driver runs it, smoke(), the default bench line.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R; T=${1:-r06_x}   # (run again as r06_y on the round's last commit)
timeout 1500 python -m pytest tests -x -q -m gpu --durations=5 > $O/${T}_pytest.txt 2>&1; tail -2 $O/${T}_pytest.txt
( time timeout 200 python __graft_entry__.py smoke ) 2>&1 | grep -v "^$\|user\|sys" | tee -a $O/${T}_pytest.txt
( time timeout 900 python bench.py > $O/${T}_bench_default.json 2> $O/${T}_bench_default.err ) 2>&1 | grep real
T=$T python - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/'+__import__("os").environ.get("T","r06_x")+'_bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['parity_crop']['pixels_within_tol'], d['secondary']['textured_leafmask']['value'])
P
