#!/bin/bash
# build_variant.sh <out.so> [-DNAME=VALUE ...]  -- an alternative build of the library for A/B runs (L3D_LIBRARY=<out.so>)
set -e
OUT=$1; shift
cd "$(dirname "$0")/../line3d_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
    -Wall -Wno-unused-function -pthread "$@" -x hip -shared -o "$OUT" l3d_kernels.hip l3d_verify_window.hip l3d_capi.hip l3d_rdd.hip l3d_affinity.hip l3d_chain.hip l3d_chain_sharded.hip line3d_host.cpp l3d_sfm.cpp l3d_segcache.cpp
