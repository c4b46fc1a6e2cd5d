#!/bin/bash
# usage (GPU box): tools/exp/job_final.sh <tag>: everything the round's profiles/ entries come from
tag=$1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_$tag.log 2>&1; tail -2 gpurun_out/gpu_tests_$tag.log
timeout 400 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || tail -5 gpurun_out/bench_$tag.err
tools/prof.sh $tag > /dev/null 2>&1
db=$(find /tmp/prof_$tag -name "*.db" | head -1); python tools/rocpd_gaps.py $db gpurun_out/gaps_$tag.txt > /dev/null
tools/pmc.sh $tag > /dev/null 2>&1
timeout 600 python bench.py --workload search --db-genomes 65000 --queries 1000 --steps 3 --warmup 1 > gpurun_out/search_$tag.json 2> gpurun_out/search_$tag.err || tail -3 gpurun_out/search_$tag.err
timeout 300 python bench.py --force-dist --cpu-clades 0 > gpurun_out/bench_fd_$tag.json 2> gpurun_out/bench_fd_$tag.err
for c in 30 70 200; do timeout 300 python bench.py --c $c --cpu-clades 0 > gpurun_out/bench_c${c}_$tag.json 2>/dev/null; done
timeout 300 python bench.py --genomes-per-gpu 5000 --cpu-clades 0 --steps 2 > gpurun_out/bench_n5000_$tag.json 2>/dev/null
timeout 300 python bench.py --clade 1000 --cpu-clades 0 --steps 1 --warmup 1 > gpurun_out/bench_dense_$tag.json 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/*_$tag.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], round(d["ms_per_step"], 3), round(d["value"]), d.get("phase_ms_per_step"))
    except Exception as e: print(f, "unreadable", e)
PY
