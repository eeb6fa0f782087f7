#!/bin/bash
# round 5, GPU call ac (the last 80 seconds of the round's budget): which test stops making progress with the early light pick (profiles/r05_aa_light_pick_issued_early.txt)?  The tail of the
# suite on the variant library under pytest-timeout (thread method: a stuck worker is killed and xdist names the test it was running).
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
PBRT_AMD_DEVICE_LIB=$R/pbrt-v3-distributed_amd/lib/variants/pick1.so timeout 58 python -m pytest tests -m gpu -q -n 4 -p no:cacheprovider --timeout 12 --timeout-method thread --durations 8 \
  -k "bssrdf or subsurface or tile_serial or maxmindist or grid_media or c5_regime or sharing" > $O/r05_ac_pytest.txt 2>&1; echo "rc $?"; grep -v "^$" $O/r05_ac_pytest.txt | tail -25 | cut -c1-220
