#!/bin/bash
# round 4, GPU run 32: the full GPU suite on the committed code once more; a kernel trace of the search workload
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_r4end.log 2>&1; grep -v "^Hostname\|^Librccl\|^RCCL\|^HIP\|^ROCm" gpurun_out/gpu_tests_r4end.log | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -1
timeout 300 python bench.py --steps 20 > gpurun_out/bench_r4end.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/bench_r4end.json')); print(round(d['ms_per_step'],3), round(d['value']/1e6,2), d['roofline']['traffic'] is not None, d['roofline_chain']['traffic'] is not None, d['cpu_baseline']['delta_vs_oracle']['max_abs_d_ani'])"
tools/prof.sh r4search --workload search --db-genomes 65000 --queries 1000 > /dev/null 2>&1; grep -E "skh::|rocprim" gpurun_out/trace_r4search.txt | head -24 | cut -c1-64,98-125
