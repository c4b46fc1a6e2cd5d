#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> [bench args...]  -> gpurun_out/trace_<tag>.md (skh:: kernels only, per step)
set -e
tag=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace -d /tmp/prof_$tag -- python $R/bench.py --steps 2 --warmup 1 --cpu-clades 0 --no-units --no-variants "$@" > /tmp/prof_$tag.log 2>&1
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $db $R/gpurun_out/trace_$tag.md "rocprofv3 --kernel-trace -- python bench.py --steps 2 --warmup 1 --cpu-clades 0 $*"
grep -E "skh::|rocprim.*trampoline|total kernel" $R/gpurun_out/trace_$tag.md | awk -F'|' '{printf "%-70s %6s %10s %10s\n", substr($2,1,70), $3, $4, $5}' > $R/gpurun_out/trace_$tag.txt
