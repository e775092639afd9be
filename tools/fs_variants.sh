#!/bin/bash
# Variant builds of one source (SRC=ffn_stream by default; the other objects come from the default build):
#   [SRC=spk_stream] tools/fs_variants.sh name="-Dflags" ...
# -> fs-eend_amd/csrc/variants/libeend_hip_<name>.so, selected with EEND_HIP_LIB.
cd "$(dirname "$0")/.." || exit 1
CS=fs-eend_amd/csrc
mkdir -p $CS/variants
SRC=${SRC:-ffn_stream}
others=$(ls $CS/*.o | grep -v "/$SRC.o")
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc $flags -c $CS/$SRC.hip -o $CS/variants/${name}_$SRC.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $CS/variants/libeend_hip_$name.so $others $CS/variants/${name}_$SRC.o &&
    rm -f $CS/variants/${name}_$SRC.o && echo "built $name" ) &
done
wait
