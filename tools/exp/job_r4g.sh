#!/bin/bash
# round 4, GPU run 31: the 512 class of the selection kernel with 384 accepted slots (9 KB instead of 12 KB of LDS per pair): A/B, the hand-over kernel's time
mkdir -p gpurun_out
cp skani_amd/libskani_hip.so /tmp/lib_keep.so
for v in acc_old acc_new acc_old acc_new acc_old acc_new; do
  cp tools/exp/variants/$v.so skani_amd/libskani_hip.so
  timeout 300 python bench.py --no-e2e --cpu-clades 0 --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()})"
done
cp /tmp/lib_keep.so skani_amd/libskani_hip.so
timeout 300 python bench.py --no-e2e --steps 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['cpu_baseline']['delta_vs_oracle'])"
tools/prof.sh r4g --no-e2e > /dev/null 2>&1; grep -E "greedy" gpurun_out/trace_r4g.txt | cut -c1-60,98-125
for c in 30 70; do timeout 300 python bench.py --c $c --cpu-clades 0 --no-e2e --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c=$c', round(d['ms_per_step'],2))"; done
