#!/bin/bash
# round 4, GPU run 29: skh_triangle's result rows read back into pinned memory: A/B on one box
mkdir -p gpurun_out
cp skani_amd/libskani_hip.so /tmp/lib_keep.so
for v in pin_old pin_new pin_old pin_new pin_old pin_new; do
  cp tools/exp/variants/$v.so skani_amd/libskani_hip.so
  BENCH_STEP_TIMES=1 timeout 300 python bench.py --no-e2e --cpu-clades 0 --steps 40 2> gpurun_out/r4p2_$v.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()})"
  grep "host view" gpurun_out/r4p2_$v.err | cut -c1-100
done
cp /tmp/lib_keep.so skani_amd/libskani_hip.so
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "triangle or wide or beyond" 2>&1 | tail -2
