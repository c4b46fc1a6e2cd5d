#!/bin/bash
# tools/exp/lib_variant.sh NAME [-DFLAG ...] : the whole library compiled with extra macros, as tools/exp/variants/NAME/libskani_hip.so -- for same-box A/B runs:
# on the GPU box `tools/exp/ab_variants.sh NAME...` runs bench.py with each variant's file in place of skani_amd/libskani_hip.so (the box's tree is a scratch copy).
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
out=tools/exp/variants/$name; mkdir -p $out/obj
objs=""
for s in alloc scan sort pack_seed sketch_build screen screen_keys chain dist rccl_transport capi; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -ffp-contract=off "$@" -c skani_amd/csrc/$s.hip -o $out/obj/$s.o > /dev/null 2>&1 &
  objs="$objs $out/obj/$s.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libskani_hip.so $objs -ldl
rm -rf $out/obj
ls -la $out/libskani_hip.so
