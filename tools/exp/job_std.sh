#!/bin/bash
# usage (GPU box): tools/exp/job_std.sh <tag> [fuzz seed]: gpu tests, default bench, kernel trace, short fuzz
tag=$1; seed=${2:-4244}
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_$tag.log 2>&1; tail -3 gpurun_out/gpu_tests_$tag.log
timeout 300 python bench.py --cpu-clades 6 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || tail -5 gpurun_out/bench_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/bench_$tag.json')); print(round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, round(d['roofline']['ms_per_launch'],3), d['cpu_baseline']['delta_vs_oracle'])"
tools/prof.sh $tag > /dev/null 2>&1; head -32 gpurun_out/trace_$tag.txt
timeout 200 python tools/fuzz_parity.py 300 $seed | tail -1
timeout 200 python tools/fuzz_parity.py 60 $seed big | tail -1
