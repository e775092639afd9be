#!/bin/bash
# Variant builds of ffn_stream.hip only (the other objects come from the default build): tools/fs_variants.sh name="-Dflags" ...
# -> fs-eend_amd/csrc/variants/libeend_hip_<name>.so, selected with EEND_HIP_LIB.
cd "$(dirname "$0")/.." || exit 1
CS=fs-eend_amd/csrc
mkdir -p $CS/variants
others=$(ls $CS/*.o | grep -v "/ffn_stream.o")
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc $flags -c $CS/ffn_stream.hip -o $CS/variants/${name}_ffn_stream.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $CS/variants/libeend_hip_$name.so $others $CS/variants/${name}_ffn_stream.o &&
    rm -f $CS/variants/${name}_ffn_stream.o && echo "built $name" ) &
done
wait
