#!/bin/bash
# exercise bench.py's N>1 path on a 1-GPU box: two ranks on GPU 0, gloo instead of RCCL; the 2-rank image must equal the 1-rank one
cd /root/repo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --spp 4 --tris 1000000 --res 640 360 --backend gloo --one-device --cpu-seconds 0 2>gpurun_out/n2.err | tail -1
timeout 600 python bench.py --gpus 1 --steps 1 --warmup 1 --spp 4 --tris 1000000 --res 640 360 --cpu-seconds 0 2>gpurun_out/n1.err | tail -1
tail -3 gpurun_out/n2.err
