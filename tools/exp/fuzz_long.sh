#!/bin/bash
# usage (GPU box, repo root): tools/exp/fuzz_long.sh <tag>  -> gpurun_out/<tag>_fuzz_long.txt : the randomised differential test at length, also with the round's A/B switches set
tag=$1
S=$(cat skani_amd/csrc/*.hip skani_amd/csrc/*.h | sha256sum | cut -c1-16)
{ echo "sources $S, $(date -u +%FT%TZ)"
  echo "tools/fuzz_parity.py 3000 60601:"; timeout 1500 python tools/fuzz_parity.py 3000 60601 2>&1 | tail -1
  echo "tools/fuzz_parity.py 60 60602 big:"; timeout 800 python tools/fuzz_parity.py 60 60602 big 2>&1 | tail -1
  for e in "SKH_TUNE_SCREEN_COL_ORDER=2" "SKH_TUNE_SCREEN_COUNT_ROWS=0" "SKH_TUNE_MARKER_GATE=0"; do echo "$e tools/fuzz_parity.py 400 60603:"; env $e timeout 600 python tools/fuzz_parity.py 400 60603 2>&1 | tail -1; done
} > gpurun_out/${tag}_fuzz_long.txt 2>&1
cat gpurun_out/${tag}_fuzz_long.txt
