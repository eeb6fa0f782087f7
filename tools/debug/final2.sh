#!/bin/bash
# short round-end check when GPU minutes are scarce: parity tests, smoke, the default bench line, the textured C3 variant
R=/root/repo; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee gpurun_out/final_pytest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/final_smoke.txt
timeout 300 python bench.py --cpu-seconds 8 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; tail -c 1200 gpurun_out/final_bench.json
timeout 300 python bench.py --textured --cpu-seconds 8 > gpurun_out/final_bench_tex.json 2> gpurun_out/final_bench_tex.err; tail -c 1500 gpurun_out/final_bench_tex.json; tail -5 gpurun_out/final_bench_tex.err
