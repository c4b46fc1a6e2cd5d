#!/bin/bash
# round 4, closing run on the final code: full GPU suite, counters (the traffic files' stamps), the bench line, kernel trace + timeline
mkdir -p gpurun_out
tag=r4f
echo "== gpu suite"; date
timeout 2400 python -m pytest tests -m gpu -x -q --durations=4 > gpurun_out/gpu_tests_$tag.log 2>&1; grep -v "^Hostname\|^Librccl\|^RCCL\|^HIP\|^ROCm" gpurun_out/gpu_tests_$tag.log | tail -8
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -1
echo "== counters"; date
tools/pmc.sh $tag > gpurun_out/pmc_$tag.log 2>&1; tail -2 gpurun_out/pmc_$tag.log | cut -c1-160
timeout 300 python bench.py --no-e2e --cpu-clades 0 > gpurun_out/bench_pre_$tag.json 2>/dev/null
python tools/make_seed_traffic.py gpurun_out/pmc_$tag.json "profiles/r04_pmc.json (tools/pmc.sh, round 4 final code)" && python tools/make_chain_traffic.py gpurun_out/pmc_$tag.json gpurun_out/bench_pre_$tag.json "profiles/r04_pmc.json (tools/pmc.sh, round 4 final code)" | tail -2
cp profiles/seed_traffic.json gpurun_out/seed_traffic_$tag.json; cp profiles/chain_traffic.json gpurun_out/chain_traffic_$tag.json
echo "== bench"; date
timeout 900 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || tail -5 gpurun_out/bench_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/bench_$tag.json')); print(round(d['ms_per_step'],3), round(d['value']/1e6,2), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, d['roofline']['frac'], d['roofline']['valu_frac'], d['roofline']['ms_per_launch'], d['roofline']['traffic'], d['roofline_chain']['traffic'], d['cpu_baseline']['delta_vs_oracle'], round(d['cpu_baseline']['value']), d['cpu_baseline']['cores'], d['e2e']['wall_s'])"
echo "== trace"; date
tools/prof.sh $tag --no-e2e > /dev/null 2>&1; head -8 gpurun_out/trace_$tag.txt | cut -c1-66,98-125
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python tools/rocpd_gaps.py $db gpurun_out/gaps_$tag.txt | head -2
python tools/rocpd_timeline.py $db gpurun_out/timeline_$tag.txt > /dev/null
echo "== force dist / config 4 / search (final code)"; date
timeout 300 python bench.py --force-dist --cpu-clades 0 --no-e2e --steps 20 > gpurun_out/${tag}_fd.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/${tag}_fd.json')); print('force-dist', round(d['ms_per_step'],3))"
timeout 900 python bench.py --collection 10000 --steps 8 --warmup 4 > gpurun_out/${tag}_config4_n1.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/${tag}_config4_n1.json')); print('config4 n1', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, d['cpu_baseline']['delta_vs_oracle']['max_abs_d_ani'])"
SKH_TUNE_DIST_KEY_RANGE_W1=1 timeout 900 python bench.py --force-dist --collection 10000 --no-e2e --cpu-clades 0 --steps 8 --warmup 4 > gpurun_out/${tag}_config4_dist_w1.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/${tag}_config4_dist_w1.json')); print('config4 dist w1', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()})"
timeout 900 python bench.py --workload search --db-genomes 65000 --queries 1000 --steps 3 --warmup 1 > gpurun_out/${tag}_search_65k.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/${tag}_search_65k.json')); print('search 65k', round(d['ms_per_step'],2), d['config']['hits'], d['config']['library_live_gb'])"
for c in 30 70 200; do timeout 300 python bench.py --c $c --cpu-clades 0 --no-e2e --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c=$c', round(d['ms_per_step'],2))"; done
timeout 300 python bench.py --genomes-per-gpu 5000 --cpu-clades 0 --no-e2e --steps 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('n5000', round(d['ms_per_step'],2), round(d['value']/1e6,1))"
date
