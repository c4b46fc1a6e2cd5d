#!/bin/bash
# round 4, GPU run 18: which allocations reach the driver in a step over 10,000 genomes
mkdir -p gpurun_out
SKH_TRACE_ALLOC=1 BENCH_STEP_TIMES=1 timeout 600 python bench.py --no-e2e --cpu-clades 0 --collection 10000 --steps 2 --warmup 2 2> gpurun_out/r4r_alloc.err > gpurun_out/r4r_alloc.json
grep -c "skh alloc" gpurun_out/r4r_alloc.err; grep "host view" gpurun_out/r4r_alloc.err
grep "skh alloc" gpurun_out/r4r_alloc.err | tail -60
