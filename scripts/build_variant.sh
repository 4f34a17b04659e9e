#!/bin/bash
# build_variant.sh <out.so> [-DNAME=VALUE ...]  -- an alternative build of the library for A/B runs (L3D_LIBRARY=<out.so>)
set -e
OUT=$(realpath -m "$1"); shift
cd "$(dirname "$0")/../line3d_amd/csrc"
make -j8 OUT="$OUT" OBJDIR="build_$(basename "$OUT" .so)" EXTRA="$*"
