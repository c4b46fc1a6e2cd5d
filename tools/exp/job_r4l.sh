#!/bin/bash
# round 4, GPU run 12: chunk kernel A/B (the faster library is used for the rest), full GPU suite, fuzz, bench, kernel trace, counters, force-dist
mkdir -p gpurun_out
short() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, 'chained', d['config'].get('chained_pairs'), (d.get('cpu_baseline') or {}).get('delta_vs_oracle'))" $1; }
for v in chunk_old chunk_mid chunk_old chunk_mid chunk_old chunk_mid; do
  cp tools/exp/variants/$v.so skani_amd/libskani_hip.so
  timeout 300 python bench.py --no-e2e --cpu-clades 0 --steps 30 > gpurun_out/r4l_ab_$v.json 2> gpurun_out/r4l_ab_$v.err && short gpurun_out/r4l_ab_$v.json || tail -3 gpurun_out/r4l_ab_$v.err
  python -c "
import json; d=json.loads(open('gpurun_out/r4l_ab_$v.json').read().strip().splitlines()[-1]); open('gpurun_out/r4l_ab_$v.ms','a').write('%f\n' % d['ms_per_step'])"
done
pick=$(python -c "
m=lambda f: min(float(x) for x in open(f).read().split())
a=m('gpurun_out/r4l_ab_chunk_old.ms'); b=m('gpurun_out/r4l_ab_chunk_mid.ms'); print('chunk_old' if a < b - 0.02 else 'chunk_mid')")
echo "picked $pick"; echo $pick > gpurun_out/r4l_pick.txt
cp tools/exp/variants/$pick.so skani_amd/libskani_hip.so
tag=r4l
echo "== gpu suite"; date
timeout 2400 python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/gpu_tests_$tag.log 2>&1; tail -12 gpurun_out/gpu_tests_$tag.log
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -1
echo "== bench"; date
timeout 900 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || tail -5 gpurun_out/bench_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/bench_$tag.json')); print(round(d['ms_per_step'],3), round(d['value']/1e6,2), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, round(d['roofline']['ms_per_launch'],3), d['roofline']['frac'], d['cpu_baseline']['delta_vs_oracle'], round(d['cpu_baseline']['value']), d['cpu_baseline']['cores'], d.get('e2e'))"
echo "== trace"; date
tools/prof.sh $tag --no-e2e > /dev/null 2>&1; head -34 gpurun_out/trace_$tag.txt | cut -c1-66,98-125
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python tools/rocpd_gaps.py $db gpurun_out/gaps_$tag.txt | head -3
python tools/rocpd_timeline.py $db gpurun_out/timeline_$tag.txt > /dev/null
echo "== counters"; date
tools/pmc.sh $tag > gpurun_out/pmc_$tag.log 2>&1; tail -3 gpurun_out/pmc_$tag.log | cut -c1-200
echo "== force dist"; date
timeout 300 python bench.py --force-dist --cpu-clades 0 --no-e2e --steps 20 > gpurun_out/bench_fd_$tag.json 2> gpurun_out/bench_fd_$tag.err; python -c "
import json; d=json.load(open('gpurun_out/bench_fd_$tag.json')); print('force-dist', round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, d['per_rank'])"
echo "== fuzz"; date
timeout 400 python tools/fuzz_parity.py 1200 $RANDOM | tail -1
timeout 200 python tools/fuzz_parity.py 100 $RANDOM big | tail -1
SKH_TUNE_WIDE_SPAN=0 timeout 300 python tools/fuzz_parity.py 400 $RANDOM | tail -1
SKH_TUNE_WIDE_SPAN=120000 timeout 300 python tools/fuzz_parity.py 400 $RANDOM | tail -1
SKH_TUNE_GREEDY_BIG_MIN=2 timeout 300 python tools/fuzz_parity.py 300 $RANDOM | tail -1
date
