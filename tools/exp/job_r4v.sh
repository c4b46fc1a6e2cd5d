#!/bin/bash
# round 4, GPU run 20: the round's closing measurements on the final code
mkdir -p gpurun_out
tag=r4v
short() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, 'value', round(d['value']), {k: d['config'].get(k) for k in ('chained_pairs','hits','library_live_gb','bytes_per_seed_position','hbm_used_gb')}, (d.get('cpu_baseline') or {}).get('delta_vs_oracle'))" $1; }
echo "== gpu suite"; date
timeout 2400 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/gpu_tests_$tag.log 2>&1; grep -v "^Hostname\|^Librccl\|^RCCL\|^HIP\|^ROCm" gpurun_out/gpu_tests_$tag.log | tail -9
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -1
echo "== bench"; date
timeout 900 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || tail -5 gpurun_out/bench_$tag.err
short gpurun_out/bench_$tag.json
python -c "
import json; d=json.load(open('gpurun_out/bench_$tag.json')); print(d['roofline']['frac'], d['roofline']['valu_frac'], d['roofline']['ms_per_launch'], d['roofline']['traffic'], d['roofline_chain']['traffic'], round(d['cpu_baseline']['value']), d['cpu_baseline']['cores'], d['e2e']['wall_s'], d['e2e']['oracle']['wall_s'])"
echo "== trace"; date
tools/prof.sh $tag --no-e2e > /dev/null 2>&1; head -6 gpurun_out/trace_$tag.txt | cut -c1-66,98-125
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python tools/rocpd_gaps.py $db gpurun_out/gaps_$tag.txt | head -2
python tools/rocpd_timeline.py $db gpurun_out/timeline_$tag.txt > /dev/null
echo "== force dist"; date
timeout 300 python bench.py --force-dist --cpu-clades 0 --no-e2e --steps 20 > gpurun_out/${tag}_fd.json 2> gpurun_out/${tag}_fd.err; short gpurun_out/${tag}_fd.json
echo "== config 4 on one GPU: plain, distributed form"; date
timeout 900 python bench.py --collection 10000 --steps 6 --warmup 2 > gpurun_out/${tag}_config4_n1.json 2> gpurun_out/${tag}_config4_n1.err || tail -5 gpurun_out/${tag}_config4_n1.err
short gpurun_out/${tag}_config4_n1.json
SKH_TUNE_DIST_KEY_RANGE_W1=1 timeout 900 python bench.py --force-dist --collection 10000 --no-e2e --cpu-clades 0 --steps 6 --warmup 2 > gpurun_out/${tag}_config4_dist_w1.json 2> gpurun_out/${tag}_config4_dist_w1.err || tail -5 gpurun_out/${tag}_config4_dist_w1.err
short gpurun_out/${tag}_config4_dist_w1.json
SKH_TUNE_DIST_KEY_RANGE_W1=1 SKH_TRACE=1 timeout 600 python bench.py --force-dist --no-e2e --cpu-clades 0 --collection 10000 --steps 1 --warmup 2 2>&1 >/dev/null | grep "skh trace\] dist" | tail -11
echo "== 8 ranks, one device"; date
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --one-device --steps 3 --warmup 1 --cpu-clades 0 > gpurun_out/${tag}_8ranks.json 2> gpurun_out/${tag}_8ranks.err || tail -5 gpurun_out/${tag}_8ranks.err
python - <<PY
import json
d=json.loads(open('gpurun_out/${tag}_8ranks.json').read().strip().splitlines()[-1])
print('ms/step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}); print(json.dumps(d['per_rank']))
PY
echo "== search"; date
timeout 900 python bench.py --workload search --db-genomes 65000 --queries 1000 --steps 3 --warmup 1 > gpurun_out/${tag}_search_65k.json 2> gpurun_out/${tag}_search_65k.err && short gpurun_out/${tag}_search_65k.json || tail -5 gpurun_out/${tag}_search_65k.err
timeout 900 python bench.py --workload search --db-genomes 113000 --queries 1000 --steps 3 --warmup 1 > gpurun_out/${tag}_search_113k.json 2> gpurun_out/${tag}_search_113k.err && short gpurun_out/${tag}_search_113k.json || tail -5 gpurun_out/${tag}_search_113k.err
echo "== presets"; date
for c in 30 70 200; do timeout 300 python bench.py --c $c --cpu-clades 0 --no-e2e --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c=$c', round(d['ms_per_step'],2))"; done
timeout 300 python bench.py --genomes-per-gpu 5000 --cpu-clades 0 --no-e2e --steps 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('n5000', round(d['ms_per_step'],2), round(d['value']/1e6,1))"
timeout 300 python bench.py --clade 1000 --cpu-clades 0 --no-e2e --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dense', round(d['ms_per_step'],1), d['config']['chained_pairs'])"
echo "== fuzz"; date
timeout 400 python tools/fuzz_parity.py 800 $RANDOM | tail -1
timeout 200 python tools/fuzz_parity.py 60 $RANDOM big | tail -1
date
