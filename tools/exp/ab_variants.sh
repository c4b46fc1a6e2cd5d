#!/bin/bash
# usage (GPU box, repo root): tools/exp/ab_variants.sh <tag> <rounds> NAME...  -> gpurun_out/<tag>_variant_<NAME>_<round>.json ; "base" = the product library
tag=$1; rounds=$2; shift 2
cp skani_amd/libskani_hip.so /tmp/base_libskani_hip.so
for r in $(seq 1 $rounds); do for v in "$@"; do
  if [ "$v" == base ]; then cp /tmp/base_libskani_hip.so skani_amd/libskani_hip.so; else cp tools/exp/variants/$v/libskani_hip.so skani_amd/libskani_hip.so; fi
  timeout 600 python bench.py --cpu-clades 2 --no-e2e --strong-collection 0 --steps 20 ${BENCH_ARGS} > gpurun_out/${tag}_variant_${v}_$r.json 2> gpurun_out/${tag}_variant_${v}_$r.err || tail -3 gpurun_out/${tag}_variant_${v}_$r.err
  python - gpurun_out/${tag}_variant_${v}_$r.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-24s" % sys.argv[2], "ms/step %.3f" % d["ms_per_step"], {k: round(v, 2) for k, v in d["phase_ms_per_step"].items()}, (d.get("cpu_baseline") or {}).get("delta_vs_oracle"))
PY
done; done
cp /tmp/base_libskani_hip.so skani_amd/libskani_hip.so
