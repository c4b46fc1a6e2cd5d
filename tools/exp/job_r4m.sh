#!/bin/bash
# round 4, GPU run 13: the bench line with the regenerated traffic files, the scaling inputs, config 4 on one GPU, 8 ranks on one device, search at 65k / 113k, presets
mkdir -p gpurun_out
tag=r4m
short() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, 'value', d['value'], {k: d['config'].get(k) for k in ('chained_pairs','hits','library_live_gb','bytes_per_seed_position','hbm_used_gb')})" $1; }
echo "== multirank tests (key ranges cut at quantiles)"; date
timeout 900 python -m pytest tests/test_bench_multirank.py tests/test_multiproc_gloo.py -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; tail -2 gpurun_out/${tag}_tests.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "triangle or screen" > gpurun_out/${tag}_tests2.log 2>&1; tail -2 gpurun_out/${tag}_tests2.log
echo "== bench"; date
timeout 900 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || tail -5 gpurun_out/bench_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/bench_$tag.json')); print(round(d['ms_per_step'],3), round(d['value']/1e6,2), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, d['roofline'], d.get('roofline_chain'), d['cpu_baseline']['delta_vs_oracle'], round(d['cpu_baseline']['value']), d['cpu_baseline']['cores'])"
echo "== predicted-scaling inputs"; date
timeout 600 python tools/predict_scaling.py 10000 > gpurun_out/${tag}_predict.json 2> gpurun_out/${tag}_predict.err && cat gpurun_out/${tag}_predict.json || tail -5 gpurun_out/${tag}_predict.err
echo "== config 4 on one GPU"; date
timeout 900 python bench.py --collection 10000 --steps 5 --warmup 1 > gpurun_out/${tag}_config4_n1.json 2> gpurun_out/${tag}_config4_n1.err || tail -5 gpurun_out/${tag}_config4_n1.err
python - <<PY
import json
d=json.loads(open('gpurun_out/${tag}_config4_n1.json').read().strip().splitlines()[-1])
print('ms/step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, d['value'], d['bases_per_s_per_gpu'], d['chained_pairs_per_s_per_gpu'], d['cpu_baseline']['delta_vs_oracle'])
PY
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
for c in 30 70 200; do timeout 300 python bench.py --c $c --cpu-clades 0 --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c=$c', round(d['ms_per_step'],2))"; done
timeout 300 python bench.py --genomes-per-gpu 5000 --cpu-clades 0 --no-e2e --steps 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('n5000', round(d['ms_per_step'],2), round(d['value']/1e6,1))"
timeout 300 python bench.py --clade 1000 --cpu-clades 0 --no-e2e --steps 1 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dense', round(d['ms_per_step'],1), d['config']['chained_pairs'])"
date
