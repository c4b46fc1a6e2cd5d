#!/bin/bash
# usage (GPU box): tools/exp/pmc_one.sh <tag> <counter> [counter ...]  -> gpurun_out/pmcone_<tag>.txt : the given counters per skh:: kernel, one bench step, one pass
tag=$1; shift; R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/po_$tag; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/po_$tag -o pmc -- python $R/bench.py --steps 1 --warmup 0 --cpu-clades 0 > /tmp/po_$tag.log 2>&1; echo rc=$?
python $R/tools/rocpd_pmc_summary.py /tmp/po_$tag/pmc_results.db $R/gpurun_out/pmcone_$tag.json > /dev/null
python - <<PY
import json
d = json.load(open("$R/gpurun_out/pmcone_$tag.json"))
with open("$R/gpurun_out/pmcone_$tag.txt", "w") as f:
    for k, v in sorted(d.items(), key=lambda kv: -kv[1]["total_ns"])[:16]:
        f.write("%-60s %.3f ms  %s\n" % (k.replace("void ", "").replace("skh::", "")[:60], v["total_ns"] / 1e6, "  ".join("%s=%d" % (c, x) for c, x in sorted(v["counters"].items()))))
print(open("$R/gpurun_out/pmcone_$tag.txt").read())
PY
