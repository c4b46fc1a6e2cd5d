#!/bin/bash
# round 4, GPU run 11: chunk kernel A/B, fuzz, key-range screen parity on the GPU, inputs of the predicted scaling table
mkdir -p gpurun_out
short() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, 'chained', d['config'].get('chained_pairs'), (d.get('cpu_baseline') or {}).get('delta_vs_oracle'))" $1; }
cp skani_amd/libskani_hip.so /tmp/lib_keep.so
for v in chunk_old chunk_new chunk_old chunk_new; do
  cp tools/exp/variants/$v.so skani_amd/libskani_hip.so
  timeout 300 python bench.py --no-e2e --cpu-clades 0 --steps 20 > gpurun_out/r4k_ab_$v.json 2> gpurun_out/r4k_ab_$v.err && short gpurun_out/r4k_ab_$v.json || tail -3 gpurun_out/r4k_ab_$v.err
done
cp /tmp/lib_keep.so skani_amd/libskani_hip.so
echo "== fuzz + parity"; date
timeout 300 python tools/fuzz_parity.py 400 4451 | tail -1
timeout 200 python tools/fuzz_parity.py 80 4452 big | tail -1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r4k_tests.log 2>&1; tail -3 gpurun_out/r4k_tests.log
echo "== dense"; date
timeout 300 python bench.py --no-e2e --cpu-clades 0 --steps 3 --clade 1000 > gpurun_out/r4k_dense.json 2> gpurun_out/r4k_dense.err && short gpurun_out/r4k_dense.json
echo "== predicted-scaling inputs"; date
timeout 600 python tools/predict_scaling.py 10000 > gpurun_out/r4k_predict.json 2> gpurun_out/r4k_predict.err && cat gpurun_out/r4k_predict.json || tail -5 gpurun_out/r4k_predict.err
tools/prof.sh r4k --no-e2e > /dev/null 2>&1; grep -E "chunk_kernel|chunk_stats|greedy_fast" gpurun_out/trace_r4k.txt
date
