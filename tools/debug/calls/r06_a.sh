#!/bin/bash
# round 6 call a: where does the 2-rank job stall?  (1) round 5's bench.py / parallel.py (exported to _r05/, same device library) 10x, the first one on the cold box
# like the driver's run; (2) this round's 30x; (3) the whole GPU suite, serial, -x, as the driver runs it.
cd /root/repo; mkdir -p gpurun_out
ln -sfn ../../pbrt-v3-distributed_amd/lib _r05/pbrt-v3-distributed_amd/lib; ln -sfn ../scenes _r05/scenes
LIMIT=240 tools/debug/n2_loop.sh /root/repo/_r05 10 r05 > gpurun_out/r06_a_loop_r05.txt 2>&1
LIMIT=200 tools/debug/n2_loop.sh /root/repo 30 r06 > gpurun_out/r06_a_loop_r06.txt 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu --durations=25 > gpurun_out/r06_a_pytest.txt 2>&1
tail -3 gpurun_out/r06_a_loop_r05.txt gpurun_out/r06_a_loop_r06.txt; tail -5 gpurun_out/r06_a_pytest.txt
