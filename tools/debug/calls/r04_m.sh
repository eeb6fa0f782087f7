export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do timeout 200 python bench.py --spp 16 --steps 2 --warmup 1 --cpu-seconds 0 --traffic none 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.load(sys.stdin); print('rcp', d['value'], d['ms_per_step'], {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()}, round(d['roofline']['nodes_per_ray'],2), round(d['roofline']['tris_per_ray'],2))"; done
