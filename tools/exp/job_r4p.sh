#!/bin/bash
# round 4, GPU run 16: the GPU's own timeline of a step over 10,000 genomes (what fills the time outside the phase timers)
mkdir -p gpurun_out
tag=r4p
BENCH_STEP_TIMES=1 timeout 600 python bench.py --no-e2e --cpu-clades 0 --collection 10000 --steps 4 --warmup 1 2>&1 >/dev/null | grep "host view"
BENCH_STEP_TIMES=1 timeout 600 python bench.py --no-e2e --cpu-clades 0 --steps 10 --warmup 2 2>&1 >/dev/null | grep "host view"
tools/prof.sh ${tag} --no-e2e --collection 10000 > /dev/null 2>&1
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python tools/rocpd_gaps.py $db gpurun_out/gaps_$tag.txt | head -3
python tools/rocpd_timeline.py $db gpurun_out/timeline_$tag.txt > /dev/null
awk '$3 > 100 || NR < 3' gpurun_out/timeline_$tag.txt | cut -c1-120 | head -120
head -40 gpurun_out/gaps_$tag.txt
