#!/bin/bash
# round 4, GPU run 17: the whole step's timeline at 10,000 genomes (both seeding launches)
mkdir -p gpurun_out
tag=r4q
tools/prof.sh ${tag} --no-e2e --collection 10000 > /dev/null 2>&1
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python tools/rocpd_gaps.py $db gpurun_out/gaps_$tag.txt | head -12
python tools/rocpd_timeline.py $db gpurun_out/timeline_$tag.txt > /dev/null
awk '$3 > 60 || NR < 3' gpurun_out/timeline_$tag.txt | cut -c1-120 | head -40
