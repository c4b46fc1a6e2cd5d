#!/bin/bash
tag=$1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "global_memory or long_genome or small_budgets" > gpurun_out/sel_$tag.log 2>&1; tail -5 gpurun_out/sel_$tag.log
SKH_TRACE=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "beyond_2_gbp" > gpurun_out/giant_$tag.log 2>&1; grep -v "^\[skh trace\] seed\|host tile" gpurun_out/giant_$tag.log | tail -42
timeout 600 python bench.py --no-e2e --cpu-clades 0 2> gpurun_out/bench_$tag.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()})"
