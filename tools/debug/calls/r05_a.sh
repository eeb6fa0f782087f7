#!/bin/bash
# round 5, GPU call a: (1) the two things round 4 shipped blind -- PBRT_AMD_SSS_WALK=1 (probe chains inside the persistent traversal lanes) and PBRT_AMD_SSS_LOG=1 (tail kernel lists the
# counted hits) -- against the shipped rounds on --subsurface (16 spp A/B, then the winner at 64 spp with its pbrt_ref crop); (2) FETCH_SIZE calibrated on a known byte count in the
# traversal's own access pattern (k_gather_probe<4> over 8 GiB) and on the streaming read, + the raw TCC request counters; (3) the box's reference lines at 16 spp (C3, textured + leaf masks).
export TMPDIR=/tmp; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
run() { tag=$1; shift; env "$@" timeout 400 python bench.py $WHAT $BARGS --steps 2 --warmup 1 --cpu-port-seconds 0 --traffic none 2> $O/r05_a_$tag.err | tail -1 > $O/r05_a_bench_$tag.json
  python - <<EOF2
import json
try:
    d = json.load(open("$O/r05_a_bench_$tag.json")); t = d.get("kernel_ms_per_step", {})
    print("$tag", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in t.items()}, (d.get("cpu_baseline") or {}).get("parity_crop", {}).get("pixels_within_tol"))
except Exception as e: print("$tag", "ERR", e)
EOF2
}
WHAT=--subsurface; BARGS="--spp 16 --cpu-seconds 0"
run sss_shipped_16spp A=1
run sss_walk_16spp PBRT_AMD_SSS_WALK=1
run sss_log_16spp PBRT_AMD_SSS_LOG=1
BARGS="--cpu-seconds 10"
run sss_walk_64spp PBRT_AMD_SSS_WALK=1
run sss_log_64spp PBRT_AMD_SSS_LOG=1
WHAT=""; BARGS="--spp 16 --cpu-seconds 0"
run c3_16spp A=1
WHAT="--textured --leafmask"
run c3_texlm_16spp A=1
# (2) calibration
rocprofv3 -L > $O/r05_a_counters_list.txt 2>&1
grep -i -c "mall\|TCC_EA0_RDREQ" $O/r05_a_counters_list.txt
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/r05_a_calib_fetch -o c --output-format csv -- python $R/tools/debug/fetch_calib.py 8 > $O/r05_a_calib_fetch.txt 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum -d $O/r05_a_calib_req -o c --output-format csv -- python $R/tools/debug/fetch_calib.py 8 > $O/r05_a_calib_req.txt 2>&1)
timeout 200 python tools/debug/fetch_calib.py 8 > $O/r05_a_calib_plain.txt 2>&1
cat $O/r05_a_calib_plain.txt
python tools/debug/fetch_calib_summary.py $O/r05_a_calib_fetch 8 | tee $O/r05_a_calib_summary.txt
python tools/debug/fetch_calib_summary.py $O/r05_a_calib_req 8 | tee -a $O/r05_a_calib_summary.txt
find $O/r05_a_calib_fetch $O/r05_a_calib_req -name "*.csv" -size +2M -delete
