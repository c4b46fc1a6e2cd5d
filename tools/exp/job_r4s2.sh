#!/bin/bash
# round 4, GPU run 27: where a search step's time outside the library's timers goes (65,000-genome database, 1,000 queries)
mkdir -p gpurun_out
BENCH_STEP_TIMES=1 SKH_TRACE=2 timeout 900 python bench.py --workload search --db-genomes 65000 --queries 1000 --steps 3 --warmup 1 2> gpurun_out/r4s2.err > gpurun_out/r4s2.json
grep "host view" gpurun_out/r4s2.err
grep "skh trace" gpurun_out/r4s2.err | tail -60 | awk '{ if ($(NF-1) > 0.2) print }' | tail -40
python -c "
import json; d=json.load(open('gpurun_out/r4s2.json')); print(round(d['ms_per_step'],2), d['phase_ms_per_step'])"
