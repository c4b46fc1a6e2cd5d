#!/bin/bash
tag=$1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pack" 2>&1 | tail -3
tools/prof.sh $tag --no-e2e > /dev/null 2>&1; grep -E "pack_kernel|seed_tiles|total" gpurun_out/trace_$tag.txt
