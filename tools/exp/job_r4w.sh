#!/bin/bash
# round 4, GPU run 21: does the device finish the step's first seeding launch on time and the host's blocking wait return late, or does the device get the launch late?
# (diagnostic library tools/exp/variants/seed_poll.so: the host polls the launch's end event before the usual blocking read-back)
mkdir -p gpurun_out
cp skani_amd/libskani_hip.so /tmp/lib_keep.so
cp tools/exp/variants/seed_poll.so skani_amd/libskani_hip.so
for poll in 0 1; do
  if [ $poll == 1 ]; then export SKH_TRACE_SEED_POLL=1; fi
  SKH_TRACE=2 BENCH_STEP_TIMES=1 timeout 600 python bench.py --no-e2e --cpu-clades 0 --collection 10000 --steps 6 --warmup 2 2> gpurun_out/r4w_$poll.err > /dev/null
  echo "poll=$poll"; grep "host view" gpurun_out/r4w_$poll.err | cut -c1-120
  grep "skh trace\] seed: \(scans\|polled\)" gpurun_out/r4w_$poll.err | tail -24 | awk '{printf "%s  ", $(NF-1)} END {print ""}'
done
cp /tmp/lib_keep.so skani_amd/libskani_hip.so
