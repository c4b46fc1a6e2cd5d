#!/bin/bash
SKH_EXP_JOIN_PROF=1 python bench.py --cpu-clades 0 --steps 2 --warmup 1 2>&1 | grep "join prof" | tail -2
