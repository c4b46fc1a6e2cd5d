#!/bin/bash
# round 4, GPU run 30: randomised differential runs against the oracle on the final library (every budget / form the tests can force)
timeout 300 python tools/fuzz_parity.py 1200 $RANDOM | tail -1
timeout 200 python tools/fuzz_parity.py 100 $RANDOM big | tail -1
SKH_TUNE_SCAN_ONE_MAX=16 SKH_TUNE_SEED_SCRATCH_BYTES=20000 timeout 300 python tools/fuzz_parity.py 400 $RANDOM | tail -1
SKH_TUNE_SCAN_ONE_MAX=16 SKH_TUNE_SCAN_TWO_MAX=16 timeout 300 python tools/fuzz_parity.py 300 $RANDOM | tail -1
SKH_TUNE_WIDE_SPAN=0 timeout 300 python tools/fuzz_parity.py 400 $RANDOM | tail -1
SKH_TUNE_WIDE_SPAN=120000 timeout 300 python tools/fuzz_parity.py 300 $RANDOM | tail -1
SKH_TUNE_GREEDY_BIG_MIN=2 timeout 300 python tools/fuzz_parity.py 300 $RANDOM | tail -1
SKH_TUNE_SEED_TILE_CAP=8 timeout 300 python tools/fuzz_parity.py 300 $RANDOM | tail -1
