#!/bin/bash
# usage: tools/exp/job_step.sh <tag>: parity subset + fuzz + bench (no cpu) + trace + timeline
tag=$1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/fuzz_parity.py 300 $RANDOM | tail -1
python bench.py --cpu-clades 0 --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()})"
tools/prof.sh $tag > /dev/null 2>&1; head -24 gpurun_out/trace_$tag.txt | cut -c1-60,95-125
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python tools/rocpd_gaps.py $db gpurun_out/gaps_$tag.txt | head -3
python tools/rocpd_timeline.py $db gpurun_out/timeline_$tag.txt
