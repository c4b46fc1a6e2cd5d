#!/bin/bash
# round 4, GPU run 8: direct seeding (chained tile scan) -- first a small run under a short timeout (a chained scan that does not end would hang), then A/B, suite, fuzz
mkdir -p gpurun_out
short() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],3), 'seed kernel', round(d['roofline']['ms_per_launch'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, 'chained', d['config'].get('chained_pairs'), (d.get('cpu_baseline') or {}).get('delta_vs_oracle'))" $1; }
echo "== small first"; date
timeout 120 python bench.py --no-e2e --cpu-clades 2 --steps 3 --genomes-per-gpu 40 > gpurun_out/r4h_small.json 2> gpurun_out/r4h_small.err && short gpurun_out/r4h_small.json || { echo "SMALL RUN FAILED rc=$?"; tail -5 gpurun_out/r4h_small.err; exit 1; }
echo "== A/B"; date
for v in 0 1 0 1; do
  SKH_TUNE_SEED_DIRECT=$v timeout 200 python bench.py --no-e2e --cpu-clades 0 --steps 20 > gpurun_out/r4h_ab_direct$v.json 2> gpurun_out/r4h_ab_direct$v.err && short gpurun_out/r4h_ab_direct$v.json || { echo "rc=$?"; tail -3 gpurun_out/r4h_ab_direct$v.err; }
done
echo "== headline with oracle"; date
timeout 600 python bench.py --steps 20 --no-e2e > gpurun_out/r4h_bench.json 2> gpurun_out/r4h_bench.err && short gpurun_out/r4h_bench.json || tail -5 gpurun_out/r4h_bench.err
echo "== fuzz + suite"; date
timeout 300 python tools/fuzz_parity.py 300 4431 | tail -1
timeout 200 python tools/fuzz_parity.py 60 4432 big | tail -1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4h_tests.log 2>&1; tail -3 gpurun_out/r4h_tests.log
echo "== host view + trace"; date
BENCH_STEP_TIMES=1 timeout 300 python bench.py --no-e2e --cpu-clades 0 --steps 20 2>&1 >/dev/null | grep "host view"
tools/prof.sh r4h --no-e2e > /dev/null 2>&1; head -14 gpurun_out/trace_r4h.txt
db=$(find /tmp/prof_r4h -name "*.db" | head -1); python tools/rocpd_gaps.py $db > gpurun_out/r4h_gaps.txt 2>&1; head -8 gpurun_out/r4h_gaps.txt; python tools/rocpd_timeline.py $db > gpurun_out/r4h_timeline.txt 2>&1
date
