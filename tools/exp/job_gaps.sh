#!/bin/bash
# usage (GPU box): tools/exp/job_gaps.sh <tag>: host time between the stages of a step (SKH_TRACE=2), kernel trace, timeline and idle gaps of the last step
tag=$1
SKH_TRACE=2 python bench.py --cpu-clades 0 --steps 3 --warmup 2 2> gpurun_out/hosttrace_$tag.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()})"
awk '/seed: host tile tables/{n++} n>=5' gpurun_out/hosttrace_$tag.txt > gpurun_out/hosttrace_last_$tag.txt; cat gpurun_out/hosttrace_last_$tag.txt
tools/prof.sh $tag > /dev/null 2>&1
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python tools/rocpd_gaps.py $db gpurun_out/gaps_$tag.txt | head -16
python tools/rocpd_timeline.py $db gpurun_out/timeline_$tag.txt; wc -l gpurun_out/timeline_$tag.txt
