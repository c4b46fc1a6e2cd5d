#!/bin/bash
SKH_TRACE=2 timeout 300 python bench.py --steps 1 --warmup 1 --no-e2e --cpu-clades 0 2>&1 >/dev/null | grep "skh trace" | tail -45
