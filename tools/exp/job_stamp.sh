#!/bin/bash
# usage (GPU box): tools/exp/job_stamp.sh <tag>: the counter passes the traffic stamps come from, a kernel trace, then the bench line (run twice: the second picks up nothing new, it is the spare)
tag=$1
tools/prof.sh $tag --no-e2e > /dev/null 2>&1
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python tools/rocpd_gaps.py $db gpurun_out/gaps_$tag.txt | head -2
python tools/rocpd_timeline.py $db gpurun_out/timeline_$tag.txt
tools/pmc.sh $tag > gpurun_out/pmc_$tag.log 2>&1; tail -2 gpurun_out/pmc_$tag.log | cut -c1-160
python tools/make_seed_traffic.py gpurun_out/pmc_$tag.json "round 3 final ($tag)" > /dev/null
timeout 600 python bench.py --no-e2e --cpu-clades 0 > gpurun_out/bench_pre_$tag.json 2>/dev/null
python tools/make_chain_traffic.py gpurun_out/pmc_$tag.json gpurun_out/bench_pre_$tag.json "round 3 final ($tag)" | tail -1
cp profiles/seed_traffic.json profiles/chain_traffic.json gpurun_out/
timeout 900 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/bench_$tag.json')); print(round(d['ms_per_step'],3), d['value'], d['roofline']['traffic'], d['roofline_chain']['traffic'], d['e2e']['runs_wall_s'])"
