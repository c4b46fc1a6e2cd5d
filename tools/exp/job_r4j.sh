#!/bin/bash
# round 4, GPU run 10: the distributed screen by key range: multi-rank tests, 8 ranks on one device (per-rank screen time), force-dist
mkdir -p gpurun_out
echo "== multirank tests"; date
timeout 900 python -m pytest tests/test_bench_multirank.py tests/test_multiproc_gloo.py -m gpu -x -q > gpurun_out/r4j_tests.log 2>&1; tail -3 gpurun_out/r4j_tests.log
echo "== 8 ranks, one device"; date
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --one-device --steps 3 --warmup 1 --cpu-clades 0 > gpurun_out/r4j_8ranks.json 2> gpurun_out/r4j_8ranks.err || tail -5 gpurun_out/r4j_8ranks.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r4j_8ranks.json').read().strip().splitlines()[-1])
print('ms/step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}); print(json.dumps(d['per_rank']))
PY
echo "== force dist"; date
for i in 1 2; do
timeout 300 python bench.py --force-dist --no-e2e --cpu-clades 0 --steps 20 > gpurun_out/r4j_force_dist.json 2> gpurun_out/r4j_force_dist.err || tail -5 gpurun_out/r4j_force_dist.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r4j_force_dist.json').read().strip().splitlines()[-1])
print('ms/step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, d['per_rank'])
PY
done
SKH_TRACE=1 timeout 300 python bench.py --force-dist --no-e2e --cpu-clades 0 --steps 2 --warmup 1 2>&1 >/dev/null | grep "skh trace\] dist" | tail -12
date
