#!/bin/bash
mkdir -p gpurun_out
BENCH_STEP_TIMES=1 timeout 600 python bench.py --no-e2e --cpu-clades 0 --collection 10000 --steps 8 --warmup 2 2>&1 >/dev/null | grep "host view"
BENCH_STEP_TIMES=1 OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 timeout 600 python bench.py --no-e2e --cpu-clades 0 --collection 10000 --steps 8 --warmup 2 2>&1 >/dev/null | grep "host view"
BENCH_STEP_TIMES=1 timeout 600 python bench.py --no-e2e --cpu-clades 0 --steps 30 --warmup 2 2>&1 >/dev/null | grep "host view"
cat /sys/fs/cgroup/cpu.max; nproc; python -c "import torch; print(torch.get_num_threads())"
