#!/bin/bash
# texture path on the GPU box: per-scene diagnostics (tools/debug/tex.py, one process per scene, concurrently);
# `tex.sh full` also runs the GPU test suite afterwards
R=/root/repo; cd $R; mkdir -p gpurun_out
for s in tex_imagemap tex_procedural tex_noise tex_mappings tex_bump tex_alpha tex_materials; do
  (timeout 150 python tools/debug/tex.py $s > gpurun_out/tex_$s.txt 2>&1; echo "exit $?" >> gpurun_out/tex_$s.txt) &
done
wait
cat gpurun_out/tex_tex_*.txt | tee gpurun_out/tex_report.txt | grep -v "close 1.0000" | tail -60
if [ "$1" = "full" ]; then timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/tex_pytest.txt; fi
