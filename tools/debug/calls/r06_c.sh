#!/bin/bash
# round 6 call c: the noise permutation table in LDS.  (1) the whole GPU suite, serial, -x, as the driver runs it; (2) the San-Miguel-like variant (textured + alpha-masked
# leaves) with its pbrt_ref crop -- round 5: 181.9 Msamples/s; (3) the default bench line complete (live PMC passes, CPU baseline, secondary); (4) rocprofv3 kernel stats of
# the plain C3 frame and of the textured + masked one.
cd /root/repo; R=/root/repo; O=$R/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 > $O/r06_c_pytest.txt 2>&1; tail -3 $O/r06_c_pytest.txt
timeout 900 python bench.py --textured --leafmask --steps 3 --warmup 1 --traffic none --cpu-seconds 8 --cpu-port-seconds 0 --secondary off > $O/r06_c_bench_c3_textured_leafmask.json 2> $O/r06_c_bench_c3_textured_leafmask.err; tail -c 600 $O/r06_c_bench_c3_textured_leafmask.json | head -c 400; echo
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/r06_c_bench_c3.json 2> $O/r06_c_bench_c3.err; head -c 400 $O/r06_c_bench_c3.json; echo
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/r06_c_prof -o c3 --output-format csv -- python $R/bench.py --cpu-seconds 0 --traffic none --secondary off > $O/r06_c_bench_c3_under_rocprof.json 2> $O/r06_c_prof.err)
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/r06_c_prof_tex -o c3tex --output-format csv -- python $R/bench.py --textured --leafmask --cpu-seconds 0 --traffic none --secondary off > $O/r06_c_bench_c3_tex_under_rocprof.json 2> $O/r06_c_prof_tex.err)
python tools/profile_summary.py stats $O/r06_c_prof $O/r06_c_kernel_stats.csv; python tools/profile_summary.py stats $O/r06_c_prof_tex $O/r06_c_kernel_stats_textured_leafmask.csv
rm -rf $O/r06_c_prof $O/r06_c_prof_tex
