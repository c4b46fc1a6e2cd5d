#!/bin/bash
# usage (GPU box): tools/exp/job_gaps.sh <tag>: bench + kernel trace + idle gaps of the last step
tag=$1
tools/exp/job_quick.sh $tag 4244 12
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python tools/rocpd_gaps.py $db gpurun_out/gaps_$tag.txt | head -24
