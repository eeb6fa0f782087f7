#!/bin/bash
# round-2 end measurement: parity suite, smoke, the default bench line (live FETCH_SIZE pass, pbrt_ref baseline), rocprofv3 kernel stats of the same
# command, the volpath / textured / C2 / C4 lines
R=/root/repo; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee gpurun_out/r02f_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r02f_smoke.txt
timeout 900 python bench.py --save-traffic > gpurun_out/r02f_bench.json 2> gpurun_out/r02f_bench.err; tail -c 3600 gpurun_out/r02f_bench.json; cp profiles/traffic_closest.json gpurun_out/r02f_traffic_closest.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02f_prof -o p --output-format csv -- python $R/bench.py --cpu-seconds 0 --cpu-port-seconds 0 --traffic none > $R/gpurun_out/r02f_bench_rocprof.json 2> $R/gpurun_out/r02f_prof.err)
find gpurun_out/r02f_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02f_kernel_stats.csv; head -12 gpurun_out/r02f_kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/r02f_prof
timeout 400 python bench.py --volpath --cpu-seconds 12 --cpu-port-seconds 0 --traffic none > gpurun_out/r02f_bench_volpath.json 2> gpurun_out/r02f_bench_volpath.err; python -c "
import json; d=json.load(open('gpurun_out/r02f_bench_volpath.json')); print('volpath 64spp', d['value'], d['kernel_ms_per_step'], d['cpu_baseline']['value'], d['cpu_baseline']['parity_crop'])"
timeout 500 python bench.py --textured --steps 2 --cpu-seconds 10 --cpu-port-seconds 0 --traffic none > gpurun_out/r02f_bench_textured.json 2> gpurun_out/r02f_bench_textured.err; python -c "
import json; d=json.load(open('gpurun_out/r02f_bench_textured.json')); print('textured 64spp', d['value'], d['kernel_ms_per_step'])"
timeout 400 python bench.py --config c2 --cpu-seconds 8 --cpu-port-seconds 0 --traffic none > gpurun_out/r02f_bench_c2.json 2> gpurun_out/r02f_bench_c2.err; python -c "
import json; d=json.load(open('gpurun_out/r02f_bench_c2.json')); print('c2', d['value'], d['kernel_ms_per_step'], d['cpu_baseline']['value'])"
timeout 500 python bench.py --config c4 --steps 2 --cpu-seconds 8 --cpu-port-seconds 0 --traffic none > gpurun_out/r02f_bench_c4.json 2> gpurun_out/r02f_bench_c4.err; python -c "
import json; d=json.load(open('gpurun_out/r02f_bench_c4.json')); print('c4', d['value'], d['kernel_ms_per_step'], d['cpu_baseline']['value'])"
