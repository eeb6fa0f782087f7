#!/bin/bash
# round 3, GPU call e: (1) VALU issue rate of partially filled waves (tools/valu_probe); (2) bench.py's N > 1 path on this 1-GPU box with the
# node-shared scene blob (two ranks on GPU 0, gloo instead of RCCL), against the 1-rank line of the same frame.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 120 tools/valu_probe/valu_probe | tee $O/r03_e_valu_probe.txt
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --spp 8 --backend gloo --one-device --cpu-seconds 0 2>$O/r03_e_n2.err | tail -1 > $O/r03_e_bench_n2_one_device_gloo.json
timeout 600 python bench.py --gpus 1 --steps 2 --warmup 1 --spp 8 --cpu-seconds 0 --traffic none 2>$O/r03_e_n1.err | tail -1 > $O/r03_e_bench_n1.json
python - <<'EOF2'
import json
for f in ("r03_e_bench_n2_one_device_gloo.json", "r03_e_bench_n1.json"):
    try:
        d = json.load(open("/root/repo/gpurun_out/" + f)); print(f, d["n_gpus"], d["value"], d["ms_per_step"], d["setup_s"])
    except Exception as e: print(f, "ERR", e)
EOF2
tail -3 $O/r03_e_n2.err
