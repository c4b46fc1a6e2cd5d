#!/bin/bash
# usage (GPU box): tools/exp/job_trace.sh <tag>: kernel trace of two bench steps, the chaining kernels' lines
tag=$1
tools/prof.sh $tag --no-e2e > /dev/null 2>&1; grep -E "chunk_stats|finalize|greedy_fast|chunk_kernel|join_count|chain_dp_thread|total" gpurun_out/trace_$tag.txt
timeout 300 python bench.py --no-e2e --cpu-clades 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()})"
