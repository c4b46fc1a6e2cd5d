#!/bin/bash
# round 4, GPU run 23: does an idle stretch behind the first step take the start-of-process transient away? (one warm-up step, then steps 2.. are timed one by one)
mkdir -p gpurun_out
for pause in 0 1.0; do
  BENCH_WARMUP_PAUSE=$pause SKH_TRACE=2 timeout 600 python bench.py --no-e2e --cpu-clades 0 --collection 10000 --steps 4 --warmup 1 2> gpurun_out/r4y_$pause.err > gpurun_out/r4y_$pause.json
  echo "pause=$pause: first launch of every step, host ms from launch to read-back:"; grep "skh trace\] seed: scans" gpurun_out/r4y_$pause.err | awk 'NR%2==1 {printf "%s  ", $(NF-1)} END {print ""}'
  python -c "
import json; d=json.loads(open('gpurun_out/r4y_$pause.json').read().strip().splitlines()[-1]); print('ms/step', round(d['ms_per_step'],2))"
done
