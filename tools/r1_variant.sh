#!/bin/bash
# Developer: build a variant of the one-row rollout kernel into its own library without touching the shipped one.
#   bash tools/r1_variant.sh NAME '-DSOCMX_R1_PLAN1D=\"RRRRRRRRXXXXXXXXLRLR\"'   ->  soc-matching_amd/socmx/libsocmx_NAME.so
# (objects other than socmx_rollout1.o are copied from the default build; load with SOCMX_LIB=...)
set -e
cd "$(dirname "$0")/../soc-matching_amd/csrc"
NAME=$1; shift
make -j8 > /dev/null
rm -rf build_$NAME; mkdir -p build_$NAME
cp -p *.o build_$NAME/
rm -f build_$NAME/socmx_rollout1.o
make OBJDIR=build_$NAME LIB=../socmx/libsocmx_$NAME.so CXXFLAGS_EXTRA="$*" 2>&1 | grep -E "error|Error" || true
ls -la ../socmx/libsocmx_$NAME.so
