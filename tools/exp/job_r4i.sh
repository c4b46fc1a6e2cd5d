#!/bin/bash
# round 4, GPU run 9: the multi-rank bench on one device (per-rank phase times for the predicted series), config 4 on one GPU, force-dist
mkdir -p gpurun_out
echo "== 8 ranks, one device"; date
timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --one-device --steps 3 --warmup 1 --cpu-clades 0 > gpurun_out/r4i_8ranks.json 2> gpurun_out/r4i_8ranks.err || tail -5 gpurun_out/r4i_8ranks.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r4i_8ranks.json').read().strip().splitlines()[-1])
print('ms/step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}); print(json.dumps(d['per_rank']))
PY
echo "== 2 ranks strong 10000"; date
timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --one-device --collection 10000 --steps 3 --warmup 1 --cpu-clades 0 > gpurun_out/r4i_2ranks_strong.json 2> gpurun_out/r4i_2ranks_strong.err || tail -5 gpurun_out/r4i_2ranks_strong.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r4i_2ranks_strong.json').read().strip().splitlines()[-1])
print('ms/step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}); print(json.dumps(d['per_rank']))
PY
echo "== config 4 on one GPU"; date
timeout 900 python bench.py --collection 10000 --steps 5 --warmup 1 > gpurun_out/r4i_config4_n1.json 2> gpurun_out/r4i_config4_n1.err || tail -5 gpurun_out/r4i_config4_n1.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r4i_config4_n1.json').read().strip().splitlines()[-1])
print('ms/step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, d['value'], d['bases_per_s_per_gpu'], d['chained_pairs_per_s_per_gpu'], d['cpu_baseline']['delta_vs_oracle'])
PY
echo "== force dist"; date
timeout 300 python bench.py --force-dist --no-e2e --cpu-clades 0 --steps 20 > gpurun_out/r4i_force_dist.json 2> gpurun_out/r4i_force_dist.err || tail -5 gpurun_out/r4i_force_dist.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r4i_force_dist.json').read().strip().splitlines()[-1])
print('ms/step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, d['per_rank'])
PY
date
