#!/bin/bash
# usage (GPU box): tools/exp/job_suite.sh <tag>: the whole GPU suite, then the default bench line
tag=$1
timeout 2400 python -m pytest tests -m gpu -x -q --durations=10 > gpurun_out/gpu_tests_$tag.log 2>&1; tail -18 gpurun_out/gpu_tests_$tag.log
timeout 900 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || tail -5 gpurun_out/bench_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/bench_$tag.json')); print(round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, d['roofline'], d['roofline_chain'], d['cpu_baseline']['delta_vs_oracle'], round(d['cpu_baseline']['value']), d['cpu_baseline']['cores'], d.get('e2e'))"
