#!/bin/bash
# tools/build_variant.sh NAME FILE.hip [-DFLAG ...] : a copy of libskani_hip.so in which ONE translation unit is replaced (another revision of a source
# file, or the same file with experiment macros), as tools/exp/variants/NAME.so -- for same-box A/B runs (tools/exp/ab.sh swaps the library files).
# FILE.hip is a path, or REV:path for `git show`.
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
python -m skani_amd.build >/dev/null 2>&1
mkdir -p tools/exp/variants /tmp/skh_variants
base=$(basename "${src##*:}" .hip)
if [[ "$src" == *:* ]]; then git show "$src" > skani_amd/csrc/_variant_$base.hip; in=skani_amd/csrc/_variant_$base.hip; else in=$src; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -ffp-contract=off "$@" -c "$in" -o /tmp/skh_variants/$name.o
rm -f skani_amd/csrc/_variant_$base.hip
objs=""
for s in alloc scan sort pack_seed sketch_build screen chain dist rccl_transport capi; do
  if [ "$s" == "$base" ]; then objs="$objs /tmp/skh_variants/$name.o"; else objs="$objs skani_amd/csrc/build/$s.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/variants/$name.so $objs -ldl
ls -la tools/exp/variants/$name.so
