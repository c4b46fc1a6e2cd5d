#!/bin/bash
# usage (GPU box): tools/exp/presets.sh <tag>  -> gpurun_out/presets_<tag>.txt : ms/step and phases for the other presets / scale points
tag=$1
out=gpurun_out/presets_$tag.txt; : > $out
run() { name=$1; shift; timeout 600 python bench.py "$@" > /tmp/p.json 2> /tmp/p.err; python - "$name" >> $out <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/p.json"))
    cb = d.get("cpu_baseline") or {}
    print(sys.argv[1], "ms/step %.2f" % d["ms_per_step"], "value %.4g %s" % (d["value"], d["unit"]), {k: round(v, 2) for k, v in d.get("phase_ms_per_step", {}).items()}, (cb.get("delta_vs_oracle") or {}))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("/tmp/p.err").read()[-400:])
PY
}
run c30 --c 30 --steps 2 --warmup 1 --cpu-clades 1
run c70 --c 70 --steps 3 --warmup 1 --cpu-clades 1
run c200 --c 200 --steps 3 --warmup 1 --cpu-clades 1
run clade1000 --clade 1000 --steps 1 --warmup 1 --cpu-clades 0
run n5000 --genomes-per-gpu 5000 --steps 2 --warmup 1 --cpu-clades 0
run search --workload search --db-genomes 65000 --queries 1000 --steps 2 --warmup 1
cat $out
