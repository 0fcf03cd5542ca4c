#!/bin/bash
# tools/ab_build.sh NAME [-DRTMI_...]... — build a variant of the native libraries into raytracing_rust_amd/lib_NAME
# (git-ignored, travels to the GPU box); select it at run time with RTMI_LIB_DIR=$PWD/raytracing_rust_amd/lib_NAME.
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export RTMI_LIB_DIR=$ROOT/raytracing_rust_amd/lib_$NAME
export RTMI_EXTRA_CFLAGS="$*"
mkdir -p $RTMI_LIB_DIR
cd $ROOT && python3 -c "
from raytracing_rust_amd import build as b
b.build_rtmi(force=True); b.build_host(force=True)
print(b.LIBRTMI)"
