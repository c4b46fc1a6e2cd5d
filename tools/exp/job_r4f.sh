#!/bin/bash
# round 4, GPU run 6: compaction A/B, where the host spends a step (SKH_TRACE=2, BENCH_STEP_TIMES), search memory with the cache trimmed
mkdir -p gpurun_out
short() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, 'chained', d['config'].get('chained_pairs'), {k: d['config'].get(k) for k in ('hits','hits_in_own_clade','hbm_used_gb','library_live_gb','bytes_per_seed_position','db_build_s')})" $1; }
cp skani_amd/libskani_hip.so /tmp/lib_keep.so
for v in compact_old compact_new compact_old compact_new; do
  cp tools/exp/variants/$v.so skani_amd/libskani_hip.so
  timeout 300 python bench.py --no-e2e --cpu-clades 0 --steps 20 > gpurun_out/r4f_ab_$v.json 2> gpurun_out/r4f_ab_$v.err && short gpurun_out/r4f_ab_$v.json || tail -3 gpurun_out/r4f_ab_$v.err
done
cp /tmp/lib_keep.so skani_amd/libskani_hip.so
echo "== host view"; date
BENCH_STEP_TIMES=1 timeout 300 python bench.py --no-e2e --cpu-clades 0 --steps 20 2>&1 >/dev/null | grep "host view"
SKH_TRACE=2 timeout 300 python bench.py --no-e2e --cpu-clades 0 --steps 3 --warmup 1 2>&1 >/dev/null | grep "skh trace" | tail -60 > gpurun_out/r4f_hosttrace.txt; cat gpurun_out/r4f_hosttrace.txt
echo "== search 65k"; date
timeout 900 python bench.py --workload search --db-genomes 65000 --queries 1000 --steps 3 --warmup 1 > gpurun_out/r4f_search_65k.json 2> gpurun_out/r4f_search_65k.err && short gpurun_out/r4f_search_65k.json || tail -5 gpurun_out/r4f_search_65k.err
date
