#!/bin/bash
# usage: tools/build_variant.sh NAME "-DFLAG=1 ..."   ->  raytrace_amd/variants/librtamd_NAME.so (select it with RT_LIB=...)
# An A/B build of the same library with extra HIP compile flags, in a build directory of its own.
set -e
name=$1; flags=$2
cd "$(dirname "$0")/.."
mkdir -p raytrace_amd/variants
make -s BUILD=build/var_$name OUT=raytrace_amd/variants/librtamd_$name.so EXTRA_HIPFLAGS="$flags" raytrace_amd/variants/librtamd_$name.so
ls -la raytrace_amd/variants/librtamd_$name.so
