#!/bin/bash
# round 4, GPU run 3: suite + fuzz on the salted tables / seeds in the join, tables beside the screen vs at sketch time, timeline + gaps
mkdir -p gpurun_out
short() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],3), 'seed kernel', round(d['roofline']['ms_per_launch'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, 'chained', d['config']['chained_pairs'], (d.get('cpu_baseline') or {}).get('delta_vs_oracle'))" $1; }
echo "== gpu suite"; date
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4c_tests.log 2>&1; tail -4 gpurun_out/r4c_tests.log
echo "== fuzz"; date
timeout 300 python tools/fuzz_parity.py 300 4411 | tail -1
timeout 200 python tools/fuzz_parity.py 60 4412 big | tail -1
echo "== A/B tables"; date
for v in at-sketch beside-screen at-sketch beside-screen; do
  timeout 300 python bench.py --no-e2e --cpu-clades 0 --steps 20 --tables $v > gpurun_out/r4c_ab_$v.json 2> gpurun_out/r4c_ab_$v.err && short gpurun_out/r4c_ab_$v.json || tail -3 gpurun_out/r4c_ab_$v.err
done
echo "== headline with oracle"; date
timeout 600 python bench.py --steps 20 --no-e2e > gpurun_out/r4c_bench.json 2> gpurun_out/r4c_bench.err && short gpurun_out/r4c_bench.json || tail -5 gpurun_out/r4c_bench.err
echo "== trace"; date
tools/prof.sh r4c --no-e2e > /dev/null 2>&1; head -24 gpurun_out/trace_r4c.txt
db=$(find /tmp/prof_r4c -name "*.db" | head -1)
python tools/rocpd_timeline.py $db > gpurun_out/r4c_timeline.txt 2>&1; python tools/rocpd_gaps.py $db > gpurun_out/r4c_gaps.txt 2>&1; head -12 gpurun_out/r4c_gaps.txt
date
