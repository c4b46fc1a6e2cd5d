#!/bin/bash
# round 4, GPU run 28: the search rows' post-processing (one stable sort, np.take): parity tests of the search path, the two database sizes
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "search or config5 or database or golden" 2>&1 | grep -v "^Hostname\|^Librccl\|^RCCL\|^HIP\|^ROCm" | tail -3
BENCH_STEP_TIMES=1 timeout 900 python bench.py --workload search --db-genomes 65000 --queries 1000 --steps 5 --warmup 1 2> gpurun_out/r4s3_65k.err > gpurun_out/r4s3_search_65k.json
grep "host view" gpurun_out/r4s3_65k.err
python -c "
import json; d=json.load(open('gpurun_out/r4s3_search_65k.json')); print('65k', round(d['ms_per_step'],2), round(d['value']), d['config']['hits'], d['config']['hits_in_own_clade'], d['phase_ms_per_step'])"
timeout 900 python bench.py --workload search --db-genomes 113000 --queries 1000 --steps 5 --warmup 1 2>/dev/null > gpurun_out/r4s3_search_113k.json
python -c "
import json; d=json.load(open('gpurun_out/r4s3_search_113k.json')); print('113k', round(d['ms_per_step'],2), round(d['value']), d['config']['hits'], d['config']['library_live_gb'], d['config']['db_build_s'])"
