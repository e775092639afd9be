#!/bin/bash
# A/B: 256-row tiles (NJ = 4) in the layer-tail stream kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
V=$PWD/fs-eend_amd/csrc/variants
EEND_FS_NJ=4 EEND_HIP_LIB=$V/libeend_hip_nj4.so timeout 300 python -m pytest tests/test_hip_ffn_stream.py -q -x -p no:cacheprovider 2>&1 | tail -1
for r in 1 2; do
echo "== default"; AB_ROUNDS=5 python tools/ab_ffn_stream.py 2>&1 | grep "stream res16"
echo "== nj4"; EEND_FS_NJ=4 EEND_HIP_LIB=$V/libeend_hip_nj4.so AB_ROUNDS=5 python tools/ab_ffn_stream.py 2>&1 | grep "stream res16"
done
EEND_FS_NJ=4 EEND_HIP_LIB=$V/libeend_hip_nj4.so timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-200
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-200
