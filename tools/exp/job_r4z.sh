#!/bin/bash
# round 4, GPU run 25: fewer launches (seeding tail in one kernel, one-launch scans, merged fills, one-launch selection order): parity, A/B against the previous library, launch count
mkdir -p gpurun_out
tag=r4z
short() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()}, 'chained', d['config'].get('chained_pairs'), (d.get('cpu_baseline') or {}).get('delta_vs_oracle'))" $1; }
echo "== parity first"; date
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/fuzz_parity.py 500 $RANDOM | tail -1
timeout 200 python tools/fuzz_parity.py 60 $RANDOM big | tail -1
SKH_TUNE_WIDE_SPAN=0 timeout 300 python tools/fuzz_parity.py 200 $RANDOM | tail -1
echo "== A/B"; date
cp skani_amd/libskani_hip.so /tmp/lib_keep.so
for v in fuse_old fuse_new fuse_old fuse_new fuse_old fuse_new; do
  cp tools/exp/variants/$v.so skani_amd/libskani_hip.so
  timeout 300 python bench.py --no-e2e --cpu-clades 0 --steps 30 > gpurun_out/${tag}_ab_$v.json 2> gpurun_out/${tag}_ab_$v.err && short gpurun_out/${tag}_ab_$v.json || tail -3 gpurun_out/${tag}_ab_$v.err
done
cp /tmp/lib_keep.so skani_amd/libskani_hip.so
echo "== headline with oracle, trace"; date
timeout 600 python bench.py --no-e2e > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; short gpurun_out/${tag}_bench.json
tools/prof.sh $tag --no-e2e > /dev/null 2>&1
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python tools/rocpd_gaps.py $db gpurun_out/gaps_$tag.txt | head -12
python tools/rocpd_timeline.py $db gpurun_out/timeline_$tag.txt > /dev/null
grep -E "seed_offsets|scan_one|fill_regions|greedy_order" gpurun_out/trace_$tag.txt | cut -c1-60,98-125
date
