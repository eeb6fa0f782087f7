#!/bin/bash
# round 4, FIRST GPU call: what round 3 built after its GPU minutes were spent (all of it verified on tools/hostemu against pbrt_ref's fixtures only) -- one call, ~12 GPU minutes:
#   1. the parity suite + smoke (the new tests: test_walked_bssrdf_probes_*, test_split_form_*, test_baseline_configs_reduced[sanmiguel_subsurface / smokebox], tile-serial samplers)
#   2. A/B of the walked BSSRDF probes:       bench.py --subsurface   vs PBRT_AMD_VOL_INLINE=1 (per-lane form), 16 spp, with the pbrt_ref crop on the walked run
#   3. A/B of the split form for grid media:  bench.py --smokebox     vs PBRT_AMD_VOL_SPLIT=0  (general form), 16 spp, with the pbrt_ref crop on the split run
#      (>= 120 Msamples/s on C3 with a grid medium -- run the 64 spp line afterwards if the 16 spp one is in range)
#   4. the tile-serial samplers at frame size: the C3 stand-in at 1080p, 4 spp, Sampler "02sequence" (launch-bound by construction: the number to beat with a hipGraph of one round)
#   5. the default C3 line (regression check of everything else: k_trace / k_shade untouched, PathState grew by two pointers)
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1000 python -m pytest tests -m gpu -x -q > $O/r04_a_pytest.txt 2>&1; tail -3 $O/r04_a_pytest.txt
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2 | tee -a $O/r04_a_pytest.txt
B="--spp 16 --steps 3 --warmup 1 --traffic none --cpu-port-seconds 0"
timeout 500 python bench.py --subsurface $B --cpu-seconds 10 2> $O/r04_a_sss_walked.err | tail -1 > $O/r04_a_bench_c3_subsurface_walked.json
PBRT_AMD_VOL_INLINE=1 timeout 500 python bench.py --subsurface $B --cpu-seconds 0 2> $O/r04_a_sss_inline.err | tail -1 > $O/r04_a_bench_c3_subsurface_per_lane.json
timeout 500 python bench.py --smokebox $B --cpu-seconds 10 2> $O/r04_a_smoke_split.err | tail -1 > $O/r04_a_bench_c3_smokebox_split.json
PBRT_AMD_VOL_SPLIT=0 timeout 500 python bench.py --smokebox $B --cpu-seconds 0 2> $O/r04_a_smoke_general.err | tail -1 > $O/r04_a_bench_c3_smokebox_general.json
# 4: the stand-in's scene file with its Sampler line swapped (the generator writes Sampler "sobol")
S=/tmp/pbrt_amd_bench/sanmiguel_synth_10000k_1920x1080_4spp
timeout 300 python bench.py --spp 4 --steps 1 --warmup 0 --traffic none --cpu-seconds 0 --cpu-port-seconds 0 > /dev/null 2>&1   # (generates the 4 spp scene directory)
sed 's/^Sampler "sobol".*$/Sampler "02sequence" "integer pixelsamples" [4]/' $S/sanmiguel_synth.pbrt > $S/sanmiguel_02seq.pbrt
timeout 600 python bench.py --scene $S/sanmiguel_02seq.pbrt --steps 1 --warmup 1 --traffic none --cpu-seconds 0 --cpu-port-seconds 0 2> $O/r04_a_02seq.err | tail -1 > $O/r04_a_bench_c3_02sequence_4spp.json
timeout 400 python bench.py --cpu-seconds 0 --traffic none 2> $O/r04_a_c3.err | tail -1 > $O/r04_a_bench_c3.json
python - <<'EOF2'
import json
for c in ("c3_subsurface_walked", "c3_subsurface_per_lane", "c3_smokebox_split", "c3_smokebox_general", "c3_02sequence_4spp", "c3"):
    try:
        d = json.load(open('/root/repo/gpurun_out/r04_a_bench_%s.json' % c))
        print(c, d['value'], d['ms_per_step'], d['kernel_ms_per_step'], (d.get('cpu_baseline') or {}).get('parity_crop', {}).get('pixels_within_tol'))
    except Exception as e: print(c, 'ERR', e)
EOF2
