#!/bin/bash
# round 2, final validation of the committed library: the whole -m gpu suite (incl. the reference's own host driving the device through the stub) + smoke
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/r02v_pytest.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a gpurun_out/r02v_pytest.txt
