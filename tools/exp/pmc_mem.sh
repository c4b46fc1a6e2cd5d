#!/bin/bash
# usage (GPU box): tools/exp/pmc_mem.sh <tag>  -> gpurun_out/pmcmem_<tag>.txt : memory-path counters (TA / TCP / TCC / UTCL1) of the join and build kernels, one bench step per pass
tag=$1; R=$PWD
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rm -rf /tmp/pm_$name; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pm_$name -o pmc -- python $R/bench.py --steps 1 --warmup 0 --cpu-clades 0 > /tmp/pm_$name.log 2>&1; echo $name rc=$?; python $R/tools/rocpd_pmc_summary.py /tmp/pm_$name/pmc_results.db $R/gpurun_out/pmcmem_${tag}_$name.json > /dev/null; }
run ta TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_TOTAL_WAVEFRONTS
run tcp TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_TOTAL_CACHE_ACCESSES
run tlb TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_TCP_TA_DATA_STALL_CYCLES
run tcc TCC_HIT TCC_MISS TCC_REQ TCC_TAG_STALL
run gr GRBM_GUI_ACTIVE TCC_BUSY TCC_CYCLE
python - <<PY
import json, glob
rows = {}
for f in sorted(glob.glob("$R/gpurun_out/pmcmem_${tag}_*.json")):
    d = json.load(open(f))
    for k, v in d.items():
        if not any(x in k for x in ("join_", "build_tables", "seed_tiles", "chunk_stats", "chain_dp_thread")): continue
        r = rows.setdefault(k.replace("void ", "").replace("skh::", "")[:40], {"ms": v["total_ns"] / 1e6})
        r.update(v["counters"])
with open("$R/gpurun_out/pmcmem_$tag.txt", "w") as out:
    for k, r in rows.items():
        out.write(k + "\n")
        for c, v in sorted(r.items()): out.write("   %-34s %s\n" % (c, ("%.3f" % v) if c == "ms" else ("%d" % v)))
print(open("$R/gpurun_out/pmcmem_$tag.txt").read())
PY
