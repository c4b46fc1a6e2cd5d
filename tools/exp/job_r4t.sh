#!/bin/bash
mkdir -p gpurun_out
SKH_TRACE=2 BENCH_STEP_TIMES=1 timeout 600 python bench.py --no-e2e --cpu-clades 0 --collection 10000 --steps 2 --warmup 2 2> gpurun_out/r4t.err > /dev/null
grep "host view" gpurun_out/r4t.err; grep "skh trace\] \(sketch\|seed\)" gpurun_out/r4t.err | tail -14
