#!/bin/bash
# the 2-rank bench job (two ranks on GPU 0, gloo) N times in a loop, each with a fresh scene directory; per iteration: rc, seconds, and -- when
# it does not finish within LIMIT s -- every process of the job gets SIGABRT (faulthandler prints all stacks) before the group is killed.
#   n2_loop.sh <root of the tree to run (bench.py there)> <iterations> <label>
ROOT=$1; N=$2; LABEL=$3; LIMIT=${LIMIT:-200}
OUT=/root/repo/gpurun_out/n2_loop_$LABEL; mkdir -p $OUT
pass=0; fail=0
for i in $(seq 1 $N); do
  D=$(mktemp -d /tmp/n2loop.XXXXXX)
  t0=$SECONDS
  PYTHONFAULTHANDLER=1 PBRT_AMD_BENCH_DIR=$D PBRT_AMD_BENCH_STACKS_S=60 setsid python $ROOT/bench.py --gpus 2 --one-device --backend gloo --steps 1 --warmup 1 --tris 200000 --res 320 192 --spp 4 --cpu-seconds 0 --traffic none >$OUT/$i.out 2>$OUT/$i.err &
  pid=$!
  waited=0
  while kill -0 $pid 2>/dev/null && [ $waited -lt $LIMIT ]; do sleep 1; waited=$((waited+1)); done
  if kill -0 $pid 2>/dev/null; then
    echo "iteration $i: STALLED after $LIMIT s -- stacks:" | tee -a $OUT/summary.txt
    ps -o pid,ppid,stat,wchan:30,etime,cmd --forest -g $(ps -o sid= -p $pid | tr -d ' ') >> $OUT/$i.err 2>&1
    for p in $(pgrep -s $(ps -o sid= -p $pid | tr -d ' ')); do kill -ABRT $p 2>/dev/null; done
    sleep 3
    kill -9 -- -$pid 2>/dev/null
    rc=stalled; fail=$((fail+1))
  else
    wait $pid; rc=$?
    if [ $rc -eq 0 ] && grep -q '^{' $OUT/$i.out; then pass=$((pass+1)); else fail=$((fail+1)); fi
  fi
    echo "iteration $i: rc $rc, $((SECONDS - t0)) s, last line: $(tail -1 $OUT/$i.err | cut -c1-160)" | tee -a $OUT/summary.txt
  rm -rf $D
  [ "$rc" = "0" ] && [ $i -gt 3 ] && rm -f $OUT/$i.err $OUT/$i.out   # keep the first three and every failure
done
echo "$LABEL: $pass passed, $fail failed of $N" | tee -a $OUT/summary.txt
ls /dev/shm | grep pbrt_amd_scene | head -5 | sed 's/^/left in \/dev\/shm: /' | tee -a $OUT/summary.txt
