#!/bin/bash
# usage (GPU box): tools/exp/job_final.sh <tag>: everything the round's profiles/ are made from
tag=$1
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/gpu_tests_$tag.log 2>&1; tail -14 gpurun_out/gpu_tests_$tag.log
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")"
timeout 900 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || tail -5 gpurun_out/bench_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/bench_$tag.json')); print(round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, round(d['roofline']['ms_per_launch'],3), d['cpu_baseline']['delta_vs_oracle'], round(d['cpu_baseline']['value']), d['cpu_baseline']['cores'], d['e2e']['wall_s'] if 'wall_s' in d.get('e2e', {}) else d.get('e2e'))"
tools/prof.sh $tag --no-e2e > /dev/null 2>&1; head -30 gpurun_out/trace_$tag.txt | cut -c1-66,98-125
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python tools/rocpd_gaps.py $db gpurun_out/gaps_$tag.txt | head -3
python tools/rocpd_timeline.py $db gpurun_out/timeline_$tag.txt
tools/pmc.sh $tag > gpurun_out/pmc_$tag.log 2>&1; tail -3 gpurun_out/pmc_$tag.log | cut -c1-200
SKH_TRACE_JOIN=1 python bench.py --cpu-clades 0 --steps 1 --warmup 1 2>&1 >/dev/null | grep "join_count" | tail -1 > gpurun_out/join_trace_$tag.txt; cat gpurun_out/join_trace_$tag.txt
timeout 300 python bench.py --force-dist --cpu-clades 0 > gpurun_out/bench_fd_$tag.json 2> gpurun_out/bench_fd_$tag.err; python -c "
import json; d=json.load(open('gpurun_out/bench_fd_$tag.json')); print('force-dist', round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()})"
timeout 600 python bench.py --workload search --db-genomes 65000 --queries 1000 --steps 3 --warmup 1 > gpurun_out/bench_search_$tag.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/bench_search_$tag.json')); print('search', round(d['ms_per_step'],2), d['config']['hits'], d['phase_ms_per_step'])"
for c in 30 70 200; do timeout 300 python bench.py --c $c --cpu-clades 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c=$c', round(d['ms_per_step'],2))"; done
timeout 300 python bench.py --genomes-per-gpu 5000 --cpu-clades 0 --steps 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('n5000', round(d['ms_per_step'],2), round(d['value']/1e6,1))"
timeout 300 python bench.py --clade 1000 --cpu-clades 0 --steps 1 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dense', round(d['ms_per_step'],1), d['config']['chained_pairs'])"
timeout 400 python tools/fuzz_parity.py 1500 $RANDOM | tail -1
timeout 200 python tools/fuzz_parity.py 100 $RANDOM big | tail -1
SKH_TUNE_WIDE_SPAN=0 timeout 300 python tools/fuzz_parity.py 600 $RANDOM | tail -1
SKH_TUNE_WIDE_SPAN=120000 timeout 300 python tools/fuzz_parity.py 600 $RANDOM | tail -1
SKH_TUNE_GREEDY_BIG_MIN=2 timeout 300 python tools/fuzz_parity.py 400 $RANDOM | tail -1
