#!/bin/bash
# round 5, GPU call y: the per-rank cost model of the N-GPU job on ONE GPU (tools/debug/scale_model.py: mi_render(rank r, world N) renders exactly rank r's tiles) on the final
# library, and the 2-rank path end to end on one device (bench.py --gpus 2 --one-device --backend gloo: sparse exchange as one batch_isend_irecv group).
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python tools/debug/scale_model.py --reps 3 2> $O/r05_y_scale.err | tail -1 | tee $O/r05_y_scale_model_c3.json
timeout 600 python bench.py --gpus 2 --one-device --backend gloo --steps 3 --warmup 1 --cpu-seconds 0 --cpu-port-seconds 0 --traffic none 2> $O/r05_y_n2.err | tail -1 | tee $O/r05_y_n2_one_device_gloo.json | cut -c1-400
