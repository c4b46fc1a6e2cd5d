#!/bin/bash
# round 4, GPU run 2: whole GPU suite + fuzz on the new seeding loop / DPP scan, A/B against the old seeding, kernel trace
mkdir -p gpurun_out
short() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],3), 'seed kernel', round(d['roofline']['ms_per_launch'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, 'chained', d['config']['chained_pairs'], (d.get('cpu_baseline') or {}).get('delta_vs_oracle'))" $1; }
echo "== gpu suite"; date
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4b_tests.log 2>&1; tail -4 gpurun_out/r4b_tests.log
echo "== fuzz"; date
timeout 300 python tools/fuzz_parity.py 300 4401 | tail -1
timeout 200 python tools/fuzz_parity.py 60 4402 big | tail -1
echo "== A/B"; date
cp skani_amd/libskani_hip.so /tmp/lib_keep.so
for v in seed_old now seed_old now; do
  if [ $v == now ]; then cp /tmp/lib_keep.so skani_amd/libskani_hip.so; else cp tools/exp/variants/$v.so skani_amd/libskani_hip.so; fi
  timeout 300 python bench.py --no-e2e --cpu-clades 0 --steps 20 > gpurun_out/r4b_ab_$v.json 2> gpurun_out/r4b_ab_$v.err && short gpurun_out/r4b_ab_$v.json || tail -3 gpurun_out/r4b_ab_$v.err
done
cp /tmp/lib_keep.so skani_amd/libskani_hip.so
echo "== headline with oracle"; date
timeout 600 python bench.py --steps 20 --no-e2e > gpurun_out/r4b_bench.json 2> gpurun_out/r4b_bench.err && short gpurun_out/r4b_bench.json || tail -5 gpurun_out/r4b_bench.err
echo "== trace"; date
tools/prof.sh r4b --no-e2e > /dev/null 2>&1; head -34 gpurun_out/trace_r4b.txt
date
