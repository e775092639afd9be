#!/bin/bash
# study builds of ffn_train_stream.hip (timing only; several are not valid kernels)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for v in base "$@"; do
  if [ $v = base ]; then unset EEND_HIP_LIB; else export EEND_HIP_LIB=$PWD/fs-eend_amd/csrc/variants/libeend_hip_$v.so; fi
  echo "== $v"; timeout 300 python tools/ab_ffn_train.py 196608 2>&1 | grep -v amdgpu.ids
done
