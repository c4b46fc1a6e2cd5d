#!/bin/bash
# usage (GPU box, repo root): tools/exp/fetch_calib.sh -> gpurun_out/fetch_calib.md
R=$PWD; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/fc_$c; timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/fc_$c -o pmc -- $R/tools/exp/fetch_calib > /tmp/fc_$c.log 2>&1
  python $R/tools/rocpd_pmc_summary.py /tmp/fc_$c/pmc_results.db $R/gpurun_out/fetch_calib_$c.json k_ > /dev/null
done
python - <<PY
import json
F = json.load(open("$R/gpurun_out/fetch_calib_FETCH_SIZE.json")); W = json.load(open("$R/gpurun_out/fetch_calib_WRITE_SIZE.json"))
GiB = 4 << 30; NR = 1 << 26
exp = {"k_stream16": (GiB, 0), "k_stream4": (GiB, 0), "k_random8": (NR * 64, 0), "k_write16": (0, GiB), "k_write4_scattered": (0, NR * 64)}
with open("$R/gpurun_out/fetch_calib.md", "w") as f:
    f.write("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE --kernel-trace -- tools/exp/fetch_calib (4 GiB buffer; counters are in KiB)\n\n")
    f.write("| kernel | ms | bytes the pattern must move (64-byte lines) | FETCH_SIZE KiB x 1024 | ratio | WRITE_SIZE KiB x 1024 | ratio |\n|---|---|---|---|---|---|---|\n")
    for k in sorted(F):
        n = k.replace("void ", ""); e = exp.get(n.split("(")[0], (0, 0))
        fs = F[k]["counters"].get("FETCH_SIZE", 0) * 1024; ws = W.get(k, {}).get("counters", {}).get("WRITE_SIZE", 0) * 1024
        f.write("| %s | %.3f | %d / %d | %d | %s | %d | %s |\n" % (n, F[k]["total_ns"] / 1e6, e[0], e[1], fs, ("%.3f" % (fs / e[0])) if e[0] else "-", ws, ("%.3f" % (ws / e[1])) if e[1] else "-"))
print(open("$R/gpurun_out/fetch_calib.md").read())
PY
