#!/bin/bash
# build_variant.sh <out.so> [-DNAME=VALUE ...]  -- an alternative build of the library for A/B runs (L3D_LIBRARY=<out.so>)
# The object directory is keyed by the flags, not by the output name: rebuilding a name with other flags never links stale objects.
set -e
OUT=$(realpath -m "$1"); shift
cd "$(dirname "$0")/../line3d_amd/csrc"
KEY=$(printf '%s' "$*" | md5sum | cut -c1-10)
make -j8 OUT="$OUT" OBJDIR="build_v_$KEY" EXTRA="$*"
