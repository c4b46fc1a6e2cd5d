#!/bin/bash
# usage (GPU box): tools/exp/job_r3a.sh <tag>: the whole -m gpu suite with durations, default bench, kernel trace
tag=$1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/gpu_tests_$tag.log 2>&1; tail -25 gpurun_out/gpu_tests_$tag.log
timeout 400 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || tail -5 gpurun_out/bench_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/bench_$tag.json')); print(round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, round(d['roofline']['ms_per_launch'],3), d['cpu_baseline']['delta_vs_oracle'], d['cpu_baseline']['seconds'])"
tools/prof.sh $tag > /dev/null 2>&1; head -40 gpurun_out/trace_$tag.txt
nproc; lscpu | grep -E "Model name|Socket|Core|Thread" 
