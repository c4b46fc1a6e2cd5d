#!/bin/bash
# usage (GPU box): tools/exp/job_wide.sh <tag>: the wide-set tests, then the default bench (the ordinary path must not have moved)
tag=$1
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --durations=8 -k "wide or beyond" > gpurun_out/wide_tests_$tag.log 2>&1; tail -25 gpurun_out/wide_tests_$tag.log
timeout 600 python bench.py --no-e2e > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || tail -5 gpurun_out/bench_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/bench_$tag.json')); print(round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, round(d['roofline']['ms_per_launch'],3), d['cpu_baseline']['delta_vs_oracle'])"
SKH_TUNE_WIDE_SPAN=0 timeout 300 python bench.py --no-e2e --cpu-clades 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('all-wide', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()})"
