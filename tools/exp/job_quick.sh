#!/bin/bash
# usage (GPU box): tools/exp/job_quick.sh <tag>: default bench (sample CPU baseline), kernel trace, short fuzz
tag=$1; seed=${2:-4244}
timeout 300 python bench.py --cpu-clades 6 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || tail -5 gpurun_out/bench_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/bench_$tag.json')); print(round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, round(d['roofline']['ms_per_launch'],3), d['cpu_baseline']['delta_vs_oracle'])"
tools/prof.sh $tag > /dev/null 2>&1; grep -v rocprim gpurun_out/trace_$tag.txt | head -${3:-14}
timeout 200 python tools/fuzz_parity.py 200 $seed | tail -1
