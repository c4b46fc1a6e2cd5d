#!/bin/bash
mkdir -p gpurun_out
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_vj
SKH_TUNE_CHAIN_ANCHORS=2000000000 rocprofv3 --kernel-trace -d /tmp/prof_vj -- python $R/tools/exp/virtual_join.py 1000 > $R/gpurun_out/r4d_vj.log 2>&1; tail -5 $R/gpurun_out/r4d_vj.log
db=$(find /tmp/prof_vj -name "*.db" | head -1)
python - <<PY
import sqlite3
db = sqlite3.connect("$db")
for r in db.execute("select name, (end-start)/1e3 from kernels where name like '%join_count%' or name like '%join_fill%' or name like '%build_tables%' order by start"):
    print("%-60s %10.1f us" % (r[0][:60], r[1]))
PY
