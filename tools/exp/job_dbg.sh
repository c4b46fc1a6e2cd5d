#!/bin/bash
SKH_TRACE=2 python bench.py --force-dist --cpu-clades 0 --steps 3 --warmup 2 2> gpurun_out/fd_trace2.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('force-dist', round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phase_ms_per_step'].items()})"
grep "skh trace" gpurun_out/fd_trace2.txt | tail -42 | head -42
