#!/bin/bash
# round 4, GPU run 7: selection with parallel decisions (A/B + fuzz + suite), host trace with entry/exit marks
mkdir -p gpurun_out
short() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, 'chained', d['config'].get('chained_pairs'), (d.get('cpu_baseline') or {}).get('delta_vs_oracle'))" $1; }
cp skani_amd/libskani_hip.so /tmp/lib_keep.so
for v in greedy_old greedy_new greedy_old greedy_new; do
  cp tools/exp/variants/$v.so skani_amd/libskani_hip.so
  timeout 300 python bench.py --no-e2e --cpu-clades 0 --steps 20 > gpurun_out/r4g_ab_$v.json 2> gpurun_out/r4g_ab_$v.err && short gpurun_out/r4g_ab_$v.json || tail -3 gpurun_out/r4g_ab_$v.err
done
cp /tmp/lib_keep.so skani_amd/libskani_hip.so
echo "== fuzz + suite"; date
timeout 300 python tools/fuzz_parity.py 400 4421 | tail -1
timeout 200 python tools/fuzz_parity.py 80 4422 big | tail -1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4g_tests.log 2>&1; tail -3 gpurun_out/r4g_tests.log
echo "== headline with oracle"; date
timeout 600 python bench.py --steps 20 --no-e2e > gpurun_out/r4g_bench.json 2> gpurun_out/r4g_bench.err && short gpurun_out/r4g_bench.json || tail -5 gpurun_out/r4g_bench.err
echo "== host view"; date
BENCH_STEP_TIMES=1 timeout 300 python bench.py --no-e2e --cpu-clades 0 --steps 20 2>&1 >/dev/null | grep "host view"
SKH_TRACE=2 timeout 300 python bench.py --no-e2e --cpu-clades 0 --steps 3 --warmup 1 2>&1 >/dev/null | grep "skh trace" | tail -28
tools/prof.sh r4g --no-e2e > /dev/null 2>&1; grep -E "greedy|seed_compact|seed_tiles" gpurun_out/trace_r4g.txt
date
