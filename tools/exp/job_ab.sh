#!/bin/bash
# A/B of the round-2 tree (tmp_r2/) against the working tree on the same box: presets and the 5000-genome collection
b() { ( cd $1 && timeout 300 python bench.py "${@:2}" --cpu-clades 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()})" ); }
for args in "--c 70" "--c 200" "--c 30" "--genomes-per-gpu 5000 --steps 3" ""; do
  echo "== $args"; echo -n "r2  "; b tmp_r2 $args; echo -n "now "; b . $args --no-e2e; echo -n "r2  "; b tmp_r2 $args; echo -n "now "; b . $args --no-e2e
done
