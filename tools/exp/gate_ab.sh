for g in 0 1 0 1; do SKH_TUNE_MARKER_GATE=$g python bench.py --cpu-clades 0 --no-e2e --strong-collection 0 --steps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('gate $g N=1000', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['phase_ms_per_step'].items()})"; done
for g in 0 1 0 1; do SKH_TUNE_MARKER_GATE=$g python bench.py --collection 10000 --steps 4 --warmup 2 --cpu-clades 0 --no-e2e 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('gate $g N=10000', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()})"; done
for g in 0 1; do SKH_TUNE_MARKER_GATE=$g python bench.py --genomes-per-gpu 5000 --steps 4 --warmup 2 --cpu-clades 0 --no-e2e --strong-collection 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('gate $g N=5000', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()})"; done
