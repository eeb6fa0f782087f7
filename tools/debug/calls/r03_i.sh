#!/bin/bash
# round 3, GPU call i: device-vs-oracle fuzz campaign on the final kernels (the quantised single-level kernels -- hot nodes in LDS, parked leaves,
# batched hand-over -- run the scenes whose instances are flattened, PBRT_AMD_INSTANCING=0; the two-level and volumetric kernels the others), the
# default bench line with the live VALU-issue figure, SQ counters of k_shade.
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
F=$O/r03_i_device_fuzz_campaign.txt; : > $F
fz() { echo "== $*" | tee -a $F; env "$@" timeout 900 python tools/fuzz_vs_reference.py --device ${ARGS} 2>&1 | tail -2 | tee -a $F; }
ARGS="--n 400 --seed 601" fz PBRT_AMD_INSTANCING=0
ARGS="--n 200 --seed 602" fz A=1
ARGS="--media --sss --n 200 --seed 603" fz A=1
ARGS="--media --n 150 --seed 604" fz PBRT_AMD_INSTANCING=0
ARGS="--spectra --n 150 --seed 605" fz PBRT_AMD_INSTANCING=0
timeout 700 python bench.py 2> $O/r03_i_c3.err | tail -1 > $O/r03_i_bench_c3.json
python - <<'EOF2'
import json
d=json.load(open('/root/repo/gpurun_out/r03_i_bench_c3.json')); r=d['roofline']
print('C3', d['value'], d['ms_per_step'], 'frac', r['frac'], 'valu_issue', r.get('valu_issue'))
EOF2
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU -d $O/r03_i_pmc_shade -o c --output-format csv -- python $R/bench.py --spp 8 --steps 1 --warmup 1 --cpu-seconds 0 --traffic none > /dev/null 2> $O/r03_i_pmc_shade.log)
python tools/profile_summary.py pmc $O/r03_i_pmc_shade $O/r03_i_pmc_shade.json > /dev/null 2>&1
python - <<'EOF2'
import json
d = json.load(open("/root/repo/gpurun_out/r03_i_pmc_shade.json"))
for k, v in d.items():
    if k.startswith("void k_shade<") or k.startswith("void k_trace<") or k.startswith("k_keycount") or k.startswith("void k_raygen"):
        r = {a: (b if a == "launches" else round(b / v["launches"])) for a, b in v.items()}
        print(k[:40], r, "lanes/VALU %.1f" % (r["SQ_THREAD_CYCLES_VALU"] / max(1, r["SQ_ACTIVE_INST_VALU"])), "VALU share of wave life %.3f" % (r["SQ_ACTIVE_INST_VALU"] / max(1, r["SQ_WAVE_CYCLES"])))
EOF2
