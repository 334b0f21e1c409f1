#!/bin/bash
# Build an A/B variant of librbl.so with extra compiler flags (e.g. -DRBL_SYM_UNROLL=4) next to the objects:
#   tools/build_variant.sh u4 -DRBL_SYM_UNROLL=4   ->  rigid_body_light_amd/build/variants/librbl_u4.so
# Select it at run time with RBL_LIBRARY=<path> (rigid_body_light_amd/_lib.py); timing comparisons must be made on ONE box.
set -e
name=$1; shift
here=$(cd "$(dirname "$0")/.." && pwd)
src=$here/rigid_body_light_amd/csrc
out=$here/rigid_body_light_amd/build/variants
mkdir -p $out/$name
for f in $(cd $src && ls rbl_*.hip) rbl_host.cpp; do
  o=$out/$name/${f%.*}.o
  if true; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -x hip -Wno-unused-function "$@" -c $src/$f -o $o &
  fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/librbl_$name.so $out/$name/*.o
echo built $out/librbl_$name.so
