#!/bin/bash
# First GPU call of round 3 (one gpurun call, ~8 min of box time):
#   1. the whole -m gpu suite -- confirms on the MI355X the stage tests that were added after round 2's last GPU call
#      (tools/hostemu only until now; they sit at the end of tests/test_gpu_parity.py);
#   2. the quad-cooperative gather probe (mi_gather_rate_coop): the request-rate ceiling of 4 x 16 B record
#      fetches per lane, plain vs quad pattern vs quad + LDS-DMA exchange, over L1 / L2 / Infinity-Cache / HBM sized buffers,
#      and whether the exchange reproduces the plain chain lane for lane;
#   3. the default bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee gpurun_out/r03a_pytest_gpu.txt
timeout 300 python - <<'PY' 2>&1 | tee gpurun_out/r03a_gather_coop.txt
import importlib, sys
sys.path.insert(0, "tests")
pa = importlib.import_module("pbrt-v3-distributed_amd")
sc = pa.Scene(text=open("scenes/cornell.pbrt").read())
ctx = pa.Context(sc, device=0)
print("%-22s %12s %12s %12s %12s   (1e9 lane requests of 16 B per second)" % ("buffer", "gather_rate4", "plain", "quad", "quad_lds"))
for name, nbytes in (("L1_32KB", 32 << 10), ("L2_2MB", 2 << 20), ("MALL_64MB", 64 << 20), ("HBM_258MB", 258 << 20), ("HBM_1GB", 1 << 30)):
    base = ctx.gather_rate(nbytes, 4)
    row = []
    for mode in (0, 1, 2):
        rate, eq, tot = ctx.gather_rate_coop(nbytes, mode)
        row.append((rate, eq, tot))
    print("%-22s %12.1f %12.1f %12.1f %12.1f   lanes equal to the plain chain: plain %d/%d, quad_lds %d/%d" %
          (name, base, row[0][0], row[1][0], row[2][0], row[0][1], row[0][2], row[2][1], row[2][2]))
ctx.close()
PY
timeout 900 python bench.py > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err; tail -c 1500 gpurun_out/r03a_bench.json
# 4. (only if the experiment build exists: `make -C pbrt-v3-distributed_amd variant NAME=coop FLAGS="-DPT_COOP_NODE_FETCH=1 -DPT_GRID_PER_CU=4"`
#    and, for the occupancy-matched baseline, `... NAME=grid4 FLAGS="-DPT_GRID_PER_CU=4"`; a third one with shorter LDS stacks so that five
#    blocks fit a CU: `... NAME=coop16 FLAGS="-DPT_COOP_NODE_FETCH=1 -DPT_LDS_STACK=16 -DPT_GRID_PER_CU=5"`)
#    the quad-cooperative node fetch inside k_trace<..., QN>: hit-level and image-level parity first, then the 16-spp C3 probe of both builds
V=$R/pbrt-v3-distributed_amd/lib/variants
if [ -f $V/coop.so ]; then
  PBRT_AMD_DEVICE_LIB=$V/coop.so timeout 600 python -m pytest tests -m gpu -x -q -k "closest_hit or render_vs_reference or baseline_configs or li_per_sample or edge_cases" 2>&1 | tail -5 | tee gpurun_out/r03a_coop_parity.txt
  for v in grid4 coop coop16; do
    [ -f $V/$v.so ] || continue
    PBRT_AMD_DEVICE_LIB=$V/$v.so timeout 300 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --cpu-port-seconds 0 --traffic none 2> gpurun_out/r03a_ab_$v.err | python -c "
import json, sys; d = json.loads(sys.stdin.read()); print('$v', d['value'], d['kernel_ms_per_step'], d['roofline'].get('request_rate', {}).get('achieved_Greq_per_s'))" | tee -a gpurun_out/r03a_ab_coop.txt
  done
fi
