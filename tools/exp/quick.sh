#!/bin/bash
# usage (GPU box): tools/exp/quick.sh <tag> [fuzz seed]  -> gpu tests, default bench, short fuzz; prints the essentials
tag=$1; seed=${2:-4244}
timeout 500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; tail -2 gpurun_out/gpu_tests.log
timeout 200 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/bench_$tag.json')); print(round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, round(d['roofline']['ms_per_launch'],3), d['cpu_baseline']['delta_vs_oracle'])"
timeout 150 python tools/fuzz_parity.py 300 $seed | tail -1
timeout 150 python tools/fuzz_parity.py 60 $seed big | tail -1
