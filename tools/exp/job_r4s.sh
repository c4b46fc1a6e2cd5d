#!/bin/bash
# round 4, GPU run 19: the allocator's larger cache: 10,000 genomes plain and distributed form, the headline, the search database at 113k (memory pressure)
mkdir -p gpurun_out
tag=r4s
SKH_TRACE_ALLOC=1 BENCH_STEP_TIMES=1 timeout 600 python bench.py --no-e2e --cpu-clades 0 --collection 10000 --steps 4 --warmup 2 2> gpurun_out/${tag}_alloc.err > gpurun_out/${tag}_plain10k.json
grep "host view" gpurun_out/${tag}_alloc.err; grep "skh alloc" gpurun_out/${tag}_alloc.err | tail -8
python -c "
import json; d=json.loads(open('gpurun_out/${tag}_plain10k.json').read().strip().splitlines()[-1]); print('10k plain', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()})"
SKH_TUNE_DIST_KEY_RANGE_W1=1 BENCH_STEP_TIMES=1 timeout 600 python bench.py --force-dist --no-e2e --cpu-clades 0 --collection 10000 --steps 4 --warmup 2 2> gpurun_out/${tag}_fd10k.err > gpurun_out/${tag}_fd10k.json
grep "host view" gpurun_out/${tag}_fd10k.err
python -c "
import json; d=json.loads(open('gpurun_out/${tag}_fd10k.json').read().strip().splitlines()[-1]); print('10k dist', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()})"
timeout 300 python bench.py --no-e2e --cpu-clades 0 --steps 30 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline', round(d['ms_per_step'],3))"
timeout 900 python bench.py --workload search --db-genomes 113000 --queries 1000 --steps 3 --warmup 1 2> gpurun_out/${tag}_search_113k.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('search 113k', round(d['ms_per_step'],2), d['config'].get('library_live_gb'), d['config'].get('hbm_used_gb'), d['config'].get('hits'))" || tail -5 gpurun_out/${tag}_search_113k.err
