#!/usr/bin/env bash
# Build oracle/_ref/libref_pair.so : the REFERENCE's own two scalar pair kernels
# (mobilityUFRPY, mobilityUFSingleWallCorrection), compiled from the source where
# it lies: /root/reference/src/c_rigid_obj.cpp (the two free functions ahead of
# `class CManyBodies`, lines 31-142).
#
# Why only these: the rest of that translation unit needs Eigen 3 and nanobind,
# neither of which exists in this image -> the reference as a whole is
# UNBUILDABLE here (DESIGN.md section "Oracle").  The two pair kernels depend on
# <cmath>/<iostream>/<stdexcept> and on `real`, which the reference selects with
# its own -DDOUBLE_PRECISION compile definition (src/eigen_defines.h:5-29); we
# compile them in double, which is the precision this project measures.
#
# The function text is piped from the reference file straight into the compiler
# through a temp file under $TMPDIR that is removed again; no reference source is
# written into this repository.  Only the resulting .so lands in oracle/_ref/
# (git-ignored; it travels to the GPU box like our own built .so files).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${RBL_REFERENCE_ROOT:-/root/reference}/src/c_rigid_obj.cpp"
OUT="$HERE/_ref"
if [ ! -f "$REF" ]; then
  echo "build_ref.sh: $REF not present (GPU box?) - keeping prebuilt $OUT" >&2
  exit 0
fi
mkdir -p "$OUT"
TMP="$(mktemp -d "${TMPDIR:-/tmp}/rbl_ref.XXXXXX")"
trap 'rm -rf "$TMP"' EXIT
START=$(grep -n '^void mobilityUFRPY' "$REF" | head -1 | cut -d: -f1)
END=$(grep -n '^class CManyBodies' "$REF" | head -1 | cut -d: -f1)
END=$((END - 1))
{
  printf '#include <cmath>\n#include <cstdlib>\n#include <iostream>\n#include <stdexcept>\n'
  printf 'using real = double; /* reference: -DDOUBLE_PRECISION, eigen_defines.h:28-29 */\n'
  sed -n "${START},${END}p" "$REF"
  cat "$HERE/ref_pair_shim.cpp"
} > "$TMP/ref_pair_tu.cpp"
g++ -O2 -ffp-contract=off -std=c++17 -fPIC -shared -o "$OUT/libref_pair.so" "$TMP/ref_pair_tu.cpp"
echo "built $OUT/libref_pair.so from $REF:${START}-${END}"
