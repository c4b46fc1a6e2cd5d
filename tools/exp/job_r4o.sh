#!/bin/bash
# round 4, GPU run 15: where the time outside the phase timers goes at 10,000 genomes (plain and distributed form, world of one); the c = 200 preset
mkdir -p gpurun_out
tag=r4o
echo "== c=200 / c=125 phases"; date
for c in 200 125; do
timeout 300 python bench.py --c $c --cpu-clades 0 --no-e2e --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c=$c', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, round(d['roofline']['ms_per_launch'],3), d['config'].get('chained_pairs'))"
done
tools/prof.sh ${tag}_c200 --no-e2e --c 200 > /dev/null 2>&1; head -24 gpurun_out/trace_${tag}_c200.txt | cut -c1-60,98-125
echo "== 10k plain: host view"; date
BENCH_STEP_TIMES=1 SKH_TRACE=2 timeout 600 python bench.py --no-e2e --cpu-clades 0 --collection 10000 --steps 2 --warmup 1 2> gpurun_out/${tag}_plain10k.err > gpurun_out/${tag}_plain10k.json
grep "host view" gpurun_out/${tag}_plain10k.err; grep "skh trace" gpurun_out/${tag}_plain10k.err | tail -70 | awk '{ if ($NF=="ms" && $(NF-1) > 0.15) print }'
echo "== 10k force-dist key range: host view"; date
SKH_TUNE_DIST_KEY_RANGE_W1=1 BENCH_STEP_TIMES=1 SKH_TRACE=2 timeout 600 python bench.py --force-dist --no-e2e --cpu-clades 0 --collection 10000 --steps 2 --warmup 1 2> gpurun_out/${tag}_fd10k.err > gpurun_out/${tag}_fd10k.json
grep "host view" gpurun_out/${tag}_fd10k.err; grep "skh trace" gpurun_out/${tag}_fd10k.err | tail -90 | awk '{ if ($NF=="ms" && $(NF-1) > 0.15) print }'
date
