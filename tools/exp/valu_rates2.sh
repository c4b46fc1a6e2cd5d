#!/bin/bash
# usage (GPU box, repo root): tools/exp/valu_rates2.sh  -> gpurun_out/valu_rates2.md (+ valu_rates2_pmc.json / .md: SQ + GRBM counters of the same kernels at W = 8)
R=$PWD
mkdir -p $R/gpurun_out
timeout 300 $R/tools/exp/valu_rates2 > $R/gpurun_out/valu_rates2.md 2>&1; echo "valu_rates2 rc=$?"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/vr2_pmc
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace -d /tmp/vr2_pmc -o pmc -- $R/tools/exp/valu_rates2 pmc > /tmp/vr2_pmc.log 2>&1; echo "pmc rc=$?"; tail -3 /tmp/vr2_pmc.log
python $R/tools/rocpd_pmc_summary.py /tmp/vr2_pmc/pmc_results.db $R/gpurun_out/valu_rates2_pmc.json k_ > /dev/null
python - <<PY
import json
d = json.load(open("$R/gpurun_out/valu_rates2_pmc.json"))
N = 65536 * 8
with open("$R/gpurun_out/valu_rates2_pmc.md", "w") as f:
    f.write("rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace -- tools/exp/valu_rates2 pmc (8 waves per SIMD)\n\n")
    f.write("| kernel | us | waves | VALU inst/wave | SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU | SQ_WAVE_CYCLES / wave / inst | SQ_BUSY_CYCLES | GRBM_GUI_ACTIVE | GRBM_COUNT | GUI_ACTIVE / us |\n|---|---|---|---|---|---|---|---|---|---|\n")
    for k, v in sorted(d.items()):
        c = v["counters"]; w = c.get("SQ_WAVES", 0) or 1; iv = c.get("SQ_INSTS_VALU", 0) or 1; us = v["total_ns"] / 1e3
        f.write("| %s | %.1f | %d | %.0f | %.3f | %.3f | %d | %d | %d | %.1f |\n" % (k.replace("void ", ""), us, w, iv / w, c.get("SQ_ACTIVE_INST_VALU", 0) / iv, c.get("SQ_WAVE_CYCLES", 0) / w / (iv / w),
                c.get("SQ_BUSY_CYCLES", 0), c.get("GRBM_GUI_ACTIVE", 0), c.get("GRBM_COUNT", 0), c.get("GRBM_GUI_ACTIVE", 0) / us))
PY
head -50 $R/gpurun_out/valu_rates2.md
