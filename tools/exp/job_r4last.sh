#!/bin/bash
# round 4, last GPU run: counters for the seeding kernel's stamp (pack_seed.hip changed: a barrier in seed_offsets_kernel), the bench line, the multi-rank bench tests
mkdir -p gpurun_out
tag=r4z9
tools/pmc.sh $tag > gpurun_out/pmc_$tag.log 2>&1; tail -1 gpurun_out/pmc_$tag.log | cut -c1-120
python tools/make_seed_traffic.py gpurun_out/pmc_$tag.json "profiles/r04_pmc.json (tools/pmc.sh, round 4 final code)" > /dev/null && echo "seed traffic regenerated"
timeout 200 python bench.py --steps 20 > gpurun_out/bench_$tag.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/bench_$tag.json')); print(round(d['ms_per_step'],3), round(d['value']/1e6,2), d['roofline']['traffic'], d['roofline_chain']['traffic'] is not None, d['cpu_baseline']['delta_vs_oracle']['max_abs_d_ani'], d['cpu_baseline']['delta_vs_oracle']['int_fields_equal'])"
timeout 200 python -m pytest tests/test_zz_bench_multirank.py -m gpu -x -q 2>&1 | grep -v "^Hostname\|^Librccl\|^RCCL\|^HIP\|^ROCm" | tail -2
