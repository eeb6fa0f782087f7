#!/bin/bash
# round 4, GPU call a: (1) tools/valu_probe/valu_probe2 -- the VALU issue ceiling in measured shader cycles; (2) the parity suite + smoke;
# (3) the default C3 line with the in-kernel shader clock (mi_trace_clock), live FETCH_SIZE and SQ passes; (4) one PMC pass with GRBM_GUI_ACTIVE / SQ_BUSY_CYCLES
# next to the kernel trace (clock = GUI-active cycles / launch duration, the guide's recipe) and one with the wave-state counters; (5) what round 3 built
# blind: bench.py --subsurface / --smokebox against their general-form partners (16 spp) and at 64 spp with the pbrt_ref crop; (6) the workload San Miguel
# really is: --textured --leafmask combined, 64 spp, pbrt_ref crop; (7) A/B of single-triangle leaves through the scene's own Accelerator parameter
# (maxnodeprims 1 vs the reference default 4): more node steps, fewer triangle tests.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 120 tools/valu_probe/valu_probe2 > $O/r04_a_valu_probe2.txt 2>&1; tail -12 $O/r04_a_valu_probe2.txt | cut -c1-200
timeout 900 python -m pytest tests -m gpu -x -q > $O/r04_a_pytest.txt 2>&1; tail -3 $O/r04_a_pytest.txt
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2 | tee -a $O/r04_a_pytest.txt
timeout 700 python bench.py --save-traffic 2> $O/r04_a_c3.err | tail -1 > $O/r04_a_bench_c3.json
cp profiles/traffic_closest.json $O/r04_a_traffic_closest.json
P="--spp 8 --steps 1 --warmup 1 --cpu-seconds 0 --traffic none"
(cd /tmp && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES --kernel-trace -d $O/r04_a_pmc_clock -o c --output-format csv -- python $R/bench.py $P > /dev/null 2> $O/r04_a_pmc_clock.log)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU -d $O/r04_a_pmc_waves -o c --output-format csv -- python $R/bench.py $P > /dev/null 2> $O/r04_a_pmc_waves.log)
B="--spp 16 --steps 2 --warmup 1 --traffic none --cpu-port-seconds 0 --cpu-seconds 0"
timeout 400 python bench.py --subsurface $B 2> $O/r04_a_sss_walked.err | tail -1 > $O/r04_a_bench_c3_subsurface_walked_16spp.json
PBRT_AMD_VOL_INLINE=1 timeout 400 python bench.py --subsurface $B 2> $O/r04_a_sss_inline.err | tail -1 > $O/r04_a_bench_c3_subsurface_per_lane_16spp.json
timeout 400 python bench.py --smokebox $B 2> $O/r04_a_smoke_split.err | tail -1 > $O/r04_a_bench_c3_smokebox_split_16spp.json
PBRT_AMD_VOL_SPLIT=0 timeout 400 python bench.py --smokebox $B 2> $O/r04_a_smoke_general.err | tail -1 > $O/r04_a_bench_c3_smokebox_general_16spp.json
F="--steps 2 --warmup 1 --traffic none --cpu-port-seconds 0 --cpu-seconds 10"
timeout 500 python bench.py --subsurface $F 2> $O/r04_a_sss64.err | tail -1 > $O/r04_a_bench_c3_subsurface.json
timeout 500 python bench.py --smokebox $F 2> $O/r04_a_smoke64.err | tail -1 > $O/r04_a_bench_c3_smokebox.json
timeout 600 python bench.py --textured --leafmask $F 2> $O/r04_a_texlm.err | tail -1 > $O/r04_a_bench_c3_textured_leafmask.json
# (7) the stand-in's own scene file with an Accelerator line
S=/tmp/pbrt_amd_bench/sanmiguel_synth_10000k_1920x1080_16spp
timeout 200 python bench.py --spp 16 --steps 1 --warmup 0 --traffic none --cpu-seconds 0 --cpu-port-seconds 0 > /dev/null 2>&1   # (generates the 16 spp scene directory)
grep -n '^Accelerator\|^WorldBegin' $S/sanmiguel_synth.pbrt | head -3
sed 's/^WorldBegin/Accelerator "bvh" "integer maxnodeprims" [1]\nWorldBegin/' $S/sanmiguel_synth.pbrt > $S/sanmiguel_leaf1.pbrt
sed 's/^WorldBegin/Accelerator "bvh" "integer maxnodeprims" [2]\nWorldBegin/' $S/sanmiguel_synth.pbrt > $S/sanmiguel_leaf2.pbrt
timeout 300 python bench.py --spp 16 --steps 2 --warmup 1 --traffic none --cpu-seconds 0 --cpu-port-seconds 0 2> $O/r04_a_leaf4.err | tail -1 > $O/r04_a_bench_c3_leaf4_16spp.json
timeout 300 python bench.py --scene $S/sanmiguel_leaf1.pbrt --steps 2 --warmup 1 --traffic none --cpu-seconds 0 --cpu-port-seconds 0 2> $O/r04_a_leaf1.err | tail -1 > $O/r04_a_bench_c3_leaf1_16spp.json
timeout 300 python bench.py --scene $S/sanmiguel_leaf2.pbrt --steps 2 --warmup 1 --traffic none --cpu-seconds 0 --cpu-port-seconds 0 2> $O/r04_a_leaf2.err | tail -1 > $O/r04_a_bench_c3_leaf2_16spp.json
python - <<'EOF2'
import json
for c in ("c3", "c3_subsurface_walked_16spp", "c3_subsurface_per_lane_16spp", "c3_smokebox_split_16spp", "c3_smokebox_general_16spp", "c3_subsurface", "c3_smokebox", "c3_textured_leafmask", "c3_leaf4_16spp", "c3_leaf1_16spp", "c3_leaf2_16spp"):
    try:
        d = json.load(open('/root/repo/gpurun_out/r04_a_bench_%s.json' % c)); r = d['roofline']
        print(c, d['value'], d['ms_per_step'], d['kernel_ms_per_step'], 'nodes/ray', round(r['nodes_per_ray'], 2), 'tris/ray', round(r['tris_per_ray'], 2), r.get('shader_clock_GHz'), (r.get('valu_issue') or {}), (d.get('cpu_baseline') or {}).get('parity_crop', {}).get('pixels_within_tol'))
    except Exception as e: print(c, 'ERR', e)
EOF2
