#!/bin/bash
cp skani_amd/libskani_hip.so /tmp/lib_keep.so
cp tools/exp/variants/pin_new.so skani_amd/libskani_hip.so
echo "library WITHOUT the barrier:"; timeout 60 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "while_the_gpu_is_busy" 2>&1 | tail -2
cp /tmp/lib_keep.so skani_amd/libskani_hip.so
echo "library with it:"; timeout 60 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "while_the_gpu_is_busy" 2>&1 | tail -2
