#!/bin/bash
# usage (GPU box): tools/exp/pmc_sq.sh <tag> [bench args]  -> gpurun_out/pmcsq_<tag>.txt : per-kernel SQ counters of one bench step (one rocprofv3 --pmc pass)
tag=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcsq_$tag
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace -d /tmp/pmcsq_$tag -o pmc -- python $R/bench.py --steps 1 --warmup 0 --cpu-clades 0 "$@" > /tmp/pmcsq_$tag.log 2>&1
python $R/tools/rocpd_pmc_summary.py /tmp/pmcsq_$tag/pmc_results.db $R/gpurun_out/pmcsq_$tag.json > /dev/null
python - <<PY
import json
d=json.load(open("$R/gpurun_out/pmcsq_$tag.json"))
rows=[]
for k,v in d.items():
    c=v["counters"]; w=c.get("SQ_WAVES",0) or 1; cyc=c.get("SQ_WAVE_CYCLES",0) or 1
    rows.append((v["total_ns"]/1e6, k[:60], v["dispatches"], int(w), c.get("SQ_INSTS_VALU",0)/w, c.get("SQ_INSTS_LDS",0)/w, 100*c.get("SQ_ACTIVE_INST_VALU",0)/cyc, 100*c.get("SQ_WAIT_INST_ANY",0)/cyc, 100*c.get("SQ_WAIT_ANY",0)/cyc, c.get("SQ_LDS_BANK_CONFLICT",0), v.get("lds_bytes",0), v.get("vgprs",0)))
rows.sort(reverse=True)
with open("$R/gpurun_out/pmcsq_$tag.txt","w") as f:
    f.write("ms kernel disp waves valu/wave lds/wave valu_busy% issue_stall% wait_any% lds_conflict lds_B vgpr\n")
    for r in rows[:24]:
        f.write("%.3f %s %d %d %.0f %.0f %.1f %.1f %.1f %d %d %d\n" % r)
PY
