#!/bin/bash
# end-to-end leg + CPU quota facts of the box
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null; nproc; python -c "import os; print(len(os.sched_getaffinity(0)))"; free -g | head -2; df -h /dev/shm | tail -1
timeout 900 python -m pytest tests/test_host_cpp.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py > gpurun_out/bench_$1.json 2> gpurun_out/bench_$1.err || tail -5 gpurun_out/bench_$1.err
python - <<PY
import json
d=json.load(open('gpurun_out/bench_$1.json'))
c=d['cpu_baseline']; print(round(d['ms_per_step'],3), round(c['value']), c['cores'], c['host'])
for r in c['sweep']: print(r['threads'], round(r['value']), {k: round(v,3) for k,v in r['seconds'].items()}, {k: round(v,2) for k,v in r['efficiency'].items()})
print(json.dumps(d.get('e2e')))
PY
