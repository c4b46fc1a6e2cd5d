#!/bin/bash
# round 4, GPU run 1: the multi-rank branch of bench.py on one device, the seeding A/B, the headline, --force-dist, config 4 on one GPU
mkdir -p gpurun_out
short() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],3), 'seed kernel', round(d['roofline']['ms_per_launch'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, 'chained', d['config']['chained_pairs'], (d.get('cpu_baseline') or {}).get('delta_vs_oracle'))" $1; }
echo "== tests"; date
timeout 2400 python -m pytest tests/test_bench_multirank.py tests/test_multiproc_gloo.py -m gpu -x -q > gpurun_out/r4a_tests.log 2>&1; tail -4 gpurun_out/r4a_tests.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "seeding or golden or fuzz or pinned" > gpurun_out/r4a_tests2.log 2>&1; tail -3 gpurun_out/r4a_tests2.log
echo "== seeding A/B"; date
cp skani_amd/libskani_hip.so /tmp/lib_keep.so
for v in seed_old seed_e64 seed_vcc seed_old seed_e64 seed_vcc; do
  cp tools/exp/variants/$v.so skani_amd/libskani_hip.so
  timeout 300 python bench.py --no-e2e --cpu-clades 0 --steps 20 > gpurun_out/r4a_ab_$v.json 2> gpurun_out/r4a_ab_$v.err && short gpurun_out/r4a_ab_$v.json || tail -3 gpurun_out/r4a_ab_$v.err
done
cp /tmp/lib_keep.so skani_amd/libskani_hip.so
echo "== headline"; date
timeout 600 python bench.py --steps 20 > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err && short gpurun_out/r4a_bench.json || tail -5 gpurun_out/r4a_bench.err
echo "== force-dist"; date
timeout 300 python bench.py --force-dist --no-e2e --cpu-clades 0 --steps 20 > gpurun_out/r4a_force_dist.json 2> gpurun_out/r4a_force_dist.err && short gpurun_out/r4a_force_dist.json || tail -5 gpurun_out/r4a_force_dist.err
echo "== config 4 on one GPU"; date
timeout 900 python bench.py --collection 10000 --steps 5 --warmup 1 > gpurun_out/r4a_config4_n1.json 2> gpurun_out/r4a_config4_n1.err && short gpurun_out/r4a_config4_n1.json || tail -5 gpurun_out/r4a_config4_n1.err
date
