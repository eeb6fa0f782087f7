#!/bin/bash
# round-end measurement: parity tests, smoke, bench (default command), rocprofv3 kernel stats + FETCH_SIZE pass, other configs
R=/root/repo; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee gpurun_out/final_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/final_smoke.txt
timeout 600 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; tail -c 1500 gpurun_out/final_bench.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final_prof -o p --output-format csv -- python $R/bench.py --cpu-seconds 0 > $R/gpurun_out/final_bench_rocprof.json 2> $R/gpurun_out/final_prof.err)
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/final_pmc -o c --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/final_bench_pmc.json 2> $R/gpurun_out/final_pmc.err)
timeout 400 python bench.py --config c2 --cpu-seconds 10 > gpurun_out/final_bench_c2.json 2> gpurun_out/final_bench_c2.err
timeout 400 python bench.py --config c4 --cpu-seconds 10 > gpurun_out/final_bench_c4.json 2> gpurun_out/final_bench_c4.err
lscpu | head -20 > gpurun_out/final_lscpu.txt
ls -la gpurun_out/final_prof gpurun_out/final_pmc 2>/dev/null | head
