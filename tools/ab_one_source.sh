#!/bin/bash
# Variant libraries that differ in ONE translation unit: tools/ab_one_source.sh <file.hip> "<name>=<-Dflags>" ...
# (the other objects are the ones of the current build; load with EEND_HIP_LIB=fs-eend_amd/csrc/variants/libeend_hip_<name>.so)
cd "$(dirname "$0")/.." || exit 1
CS=fs-eend_amd/csrc; src=$1; shift
mkdir -p $CS/variants
others=$(python -c "import sys; sys.path.insert(0, '.'); import fs_eend_amd.build as b; print(' '.join('$CS/' + s[:-4] + '.o' for s in b.SOURCES if s != '$src'))")
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc $flags -c $CS/$src -o $CS/variants/${name}_${src%.hip}.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $CS/variants/libeend_hip_$name.so $others $CS/variants/${name}_${src%.hip}.o &&
    rm -f $CS/variants/${name}_${src%.hip}.o && echo "built $CS/variants/libeend_hip_$name.so" ) &
done
wait
