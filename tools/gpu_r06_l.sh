#!/bin/bash
# round 6, call L: fragment-major f32 rows in the layer-tail stream kernel (timing study, LS decoder size)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for v in base fragmajor; do
  if [ $v = base ]; then unset EEND_HIP_LIB; else export EEND_HIP_LIB=$PWD/fs-eend_amd/csrc/variants/libeend_hip_$v.so; fi
  echo "== $v"; AB_M=327680 AB_ROUNDS=5 timeout 300 python tools/ab_ffn_stream.py 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee gpurun_out/r06_ffn_stream_fragmajor.txt
