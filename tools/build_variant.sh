#!/bin/bash
# study build: one source recompiled with extra -D flags, linked with the other objects into csrc/variants/libeend_hip_<name>.so
# usage: tools/build_variant.sh <name> <file.hip> -DFOO ...   (select with EEND_HIP_LIB=<path>)
set -e
cd /root/repo/fs-eend_amd/csrc; name=$1; src=$2; shift 2
mkdir -p variants
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc "$@" -c $src -o variants/${src%.hip}_$name.o
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
hipcc -shared -fPIC --offload-arch=gfx950 -o variants/libeend_hip_$name.so $objs variants/${src%.hip}_$name.o
echo built variants/libeend_hip_$name.so
