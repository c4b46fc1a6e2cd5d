#!/bin/bash
# round 4, GPU run 22: config 4's collection on one GPU past the start-of-process transient (profiles/r04_first_steps_transient.md): plain and distributed form
mkdir -p gpurun_out
tag=r4x
short() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, 'value', round(d['value']), (d.get('cpu_baseline') or {}).get('delta_vs_oracle'))" $1; }
BENCH_STEP_TIMES=1 timeout 900 python bench.py --collection 10000 --steps 8 --warmup 4 > gpurun_out/${tag}_config4_n1.json 2> gpurun_out/${tag}_config4_n1.err || tail -5 gpurun_out/${tag}_config4_n1.err
short gpurun_out/${tag}_config4_n1.json; grep "host view" gpurun_out/${tag}_config4_n1.err | cut -c1-110
SKH_TUNE_DIST_KEY_RANGE_W1=1 BENCH_STEP_TIMES=1 timeout 900 python bench.py --force-dist --collection 10000 --no-e2e --cpu-clades 0 --steps 8 --warmup 4 > gpurun_out/${tag}_config4_dist_w1.json 2> gpurun_out/${tag}_config4_dist_w1.err || tail -5 gpurun_out/${tag}_config4_dist_w1.err
short gpurun_out/${tag}_config4_dist_w1.json; grep "host view" gpurun_out/${tag}_config4_dist_w1.err | cut -c1-110
