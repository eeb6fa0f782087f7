#!/bin/bash
# PMC passes over a short bench run (8 spp): tools/debug/pmc.sh <tag> <counters...>; CSV lands in gpurun_out/pmc_<tag>/
export TMPDIR=/tmp; R=/root/repo; tag=$1; shift
cd /tmp && timeout 500 rocprofv3 --pmc "$@" -d $R/gpurun_out/pmc_$tag -o c --output-format csv -- python $R/bench.py --spp 8 --steps 1 --warmup 1 --cpu-seconds 0 > /dev/null 2> $R/gpurun_out/pmc_$tag.log
