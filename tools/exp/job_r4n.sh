#!/bin/bash
# round 4, GPU run 14: the cell gather on device buffers: multi-rank tests, RCCL with a world of one by key range, 8 ranks on one device, force-dist both forms
mkdir -p gpurun_out
tag=r4n
echo "== multirank tests"; date
timeout 900 python -m pytest tests/test_bench_multirank.py tests/test_multiproc_gloo.py -m gpu -x -q 2>&1 | grep -v "^Hostname\|^Librccl\|^RCCL\|^HIP\|^ROCm" | tail -4
echo "== force dist, row form / key-range form"; date
for kr in 0 1; do
SKH_TUNE_DIST_KEY_RANGE_W1=$kr timeout 300 python bench.py --force-dist --no-e2e --cpu-clades 0 --steps 20 > gpurun_out/${tag}_fd_$kr.json 2> gpurun_out/${tag}_fd_$kr.err || tail -5 gpurun_out/${tag}_fd_$kr.err
python - <<PY
import json
d=json.loads(open('gpurun_out/${tag}_fd_$kr.json').read().strip().splitlines()[-1])
print('key range $kr: ms/step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, d['per_rank'])
PY
done
SKH_TUNE_DIST_KEY_RANGE_W1=1 SKH_TRACE=1 timeout 300 python bench.py --force-dist --no-e2e --cpu-clades 0 --steps 2 --warmup 1 2>&1 >/dev/null | grep "skh trace\] dist" | tail -12
echo "== force dist at 10,000 genomes, key-range form"; date
SKH_TUNE_DIST_KEY_RANGE_W1=1 SKH_TRACE=1 timeout 600 python bench.py --force-dist --no-e2e --cpu-clades 0 --collection 10000 --steps 2 --warmup 1 2> gpurun_out/${tag}_fd10k.err > gpurun_out/${tag}_fd10k.json; grep "skh trace\] dist" gpurun_out/${tag}_fd10k.err | tail -12
python - <<PY
import json
d=json.loads(open('gpurun_out/${tag}_fd10k.json').read().strip().splitlines()[-1])
print('10k key range: ms/step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, d['per_rank'])
PY
echo "== 8 ranks, one device"; date
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --one-device --steps 3 --warmup 1 --cpu-clades 0 > gpurun_out/${tag}_8ranks.json 2> gpurun_out/${tag}_8ranks.err || tail -5 gpurun_out/${tag}_8ranks.err
python - <<PY
import json
d=json.loads(open('gpurun_out/${tag}_8ranks.json').read().strip().splitlines()[-1])
print('ms/step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}); print(json.dumps(d['per_rank']))
PY
date
