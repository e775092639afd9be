#!/bin/bash
# Same-box A/B of kernel variants: builds extra copies of the library with -D flags (CPU side, before gpurun) and
# times a study script against each via EEND_HIP_LIB.   usage: tools/ab_variants.sh build "<name>=<-Dflags>" ...
#                                                               tools/ab_variants.sh run <study.py> <name> ...
cd "$(dirname "$0")/.." || exit 1
CS=fs-eend_amd/csrc
if [ "$1" = build ]; then
  shift
  mkdir -p $CS/variants
  for spec in "$@"; do
    name=${spec%%=*}; flags=${spec#*=}
    objs=""
    for f in $(python -c "import sys; sys.path.insert(0, '.'); import fs_eend_amd.build as b; print(' '.join(s[:-4] for s in b.SOURCES))"); do
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc $flags -c $CS/$f.hip -o $CS/variants/${name}_$f.o &
      objs="$objs $CS/variants/${name}_$f.o"
    done
    wait
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $CS/variants/libeend_hip_$name.so $objs && rm -f $objs
    echo "built $CS/variants/libeend_hip_$name.so"
  done
else
  shift; study=$1; shift
  for round in 1 2; do
    echo "== default"; python $study
    for name in "$@"; do echo "== $name"; EEND_HIP_LIB=$PWD/$CS/variants/libeend_hip_$name.so python $study; done
  done
fi
