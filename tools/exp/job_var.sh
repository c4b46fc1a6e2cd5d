#!/bin/bash
# usage (GPU box): tools/exp/job_var.sh <kernel-name-pattern>: every library under tools/exp/variants in turn (twice): the step and the named kernels
pat=$1
cp skani_amd/libskani_hip.so /tmp/orig.so
for rep in 1 2; do
for lib in tools/exp/variants/libskani_hip_*.so; do
  cp $lib skani_amd/libskani_hip.so
  tag=v$(basename $lib .so | sed 's/libskani_hip_//')_$rep
  tools/prof.sh $tag --no-e2e > /dev/null 2>&1
  echo "$tag: $(grep -E "$pat" gpurun_out/trace_$tag.txt | awk '{print $(NF)}' | tr '\n' ' ') us;  $(timeout 300 python bench.py --no-e2e --cpu-clades 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step', round(d['ms_per_step'],3), 'chain', round(d['phase_ms_per_step']['chain_ms'],3))")"
done; done
cp /tmp/orig.so skani_amd/libskani_hip.so
