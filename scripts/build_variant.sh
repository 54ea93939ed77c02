#!/bin/bash
# Builds a variant of liblpx.so (diagnostic / experiment builds; never shipped):
#   scripts/build_variant.sh NAME "-DLPX_CHAIN2_FINE ..."  ->  gpurun_variants/liblpx_NAME.so   (use with LPX_LIB_PATH)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $R/gpurun_variants
make -C $R/linear_programming_solver_amd/csrc OUT=$R/gpurun_variants/liblpx_$NAME.so OBJDIR=$R/build/csrc_$NAME \
  CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result $*"
