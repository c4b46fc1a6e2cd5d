#!/bin/bash
# round 4, GPU run 5: join tile groups ordered by probed sketch (A/B), config 5 with compact shards, a 113,000-genome database
mkdir -p gpurun_out
short() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, 'chained', d['config'].get('chained_pairs'), (d.get('cpu_baseline') or {}).get('delta_vs_oracle'), {k: d['config'].get(k) for k in ('hits','hits_in_own_clade','hbm_used_gb','db_build_s')})" $1; }
cp skani_amd/libskani_hip.so /tmp/lib_keep.so
for v in join_unsorted join_sorted join_unsorted join_sorted; do
  cp tools/exp/variants/$v.so skani_amd/libskani_hip.so
  timeout 300 python bench.py --no-e2e --cpu-clades 0 --steps 20 > gpurun_out/r4e_ab_$v.json 2> gpurun_out/r4e_ab_$v.err && short gpurun_out/r4e_ab_$v.json || tail -3 gpurun_out/r4e_ab_$v.err
done
cp /tmp/lib_keep.so skani_amd/libskani_hip.so
echo "== dense + N=5000 with the sorted order"; date
timeout 300 python bench.py --no-e2e --cpu-clades 0 --steps 3 --clade 1000 > gpurun_out/r4e_dense.json 2> gpurun_out/r4e_dense.err && short gpurun_out/r4e_dense.json || tail -3 gpurun_out/r4e_dense.err
timeout 300 python bench.py --no-e2e --cpu-clades 0 --steps 3 --genomes-per-gpu 5000 > gpurun_out/r4e_n5000.json 2> gpurun_out/r4e_n5000.err && short gpurun_out/r4e_n5000.json || tail -3 gpurun_out/r4e_n5000.err
echo "== parity"; date
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "triangle or config3 or search or edge or large or fuzz" > gpurun_out/r4e_tests.log 2>&1; tail -3 gpurun_out/r4e_tests.log
echo "== search, compact shards"; date
timeout 900 python bench.py --workload search --db-genomes 65000 --queries 1000 --steps 3 --warmup 1 > gpurun_out/r4e_search_65k.json 2> gpurun_out/r4e_search_65k.err && short gpurun_out/r4e_search_65k.json || tail -5 gpurun_out/r4e_search_65k.err
timeout 900 python bench.py --workload search --db-genomes 113000 --queries 1000 --steps 3 --warmup 1 > gpurun_out/r4e_search_113k.json 2> gpurun_out/r4e_search_113k.err && short gpurun_out/r4e_search_113k.json || tail -5 gpurun_out/r4e_search_113k.err
date
