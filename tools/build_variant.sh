#!/bin/bash
# tools/build_variant.sh NAME "EXTRA_FLAGS" file.hip [file2.hip ...]
# Builds cross-resolution-face-recognition_amd/xrface/libxrface_NAME.so: the listed sources recompiled with EXTRA_FLAGS, every other
# object taken from the regular build (run `make -C cross-resolution-face-recognition_amd/csrc` first).  Select it at run time with
# XR_LIB=<path> (xrface/_lib.py) -- in-process A/B of kernel variants without touching the default library.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/cross-resolution-face-recognition_amd/csrc
NAME=$1; FLAGS=$2; shift 2
mkdir -p $CS/build_$NAME
OBJS=""
for f in $CS/build/*.o; do
  b=$(basename $f .o); skip=0
  for src in "$@"; do [ "$(basename $src .hip)" = "$b" ] && skip=1; done
  [ $skip = 0 ] && OBJS="$OBJS $f"
done
for src in "$@"; do
  b=$(basename $src .hip)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I$CS -I$ROOT/include $FLAGS -c $CS/$b.hip -o $CS/build_$NAME/$b.o &
  OBJS="$OBJS $CS/build_$NAME/$b.o"
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS -o $ROOT/cross-resolution-face-recognition_amd/xrface/libxrface_$NAME.so
echo built libxrface_$NAME.so
