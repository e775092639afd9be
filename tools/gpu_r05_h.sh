#!/bin/bash
# A/B: residual rows of all token fragments requested under the last out-projection items (ffn_stream RES16, spk_stream) vs the shipped order
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
V=$PWD/fs-eend_amd/csrc/variants
EEND_HIP_LIB=$V/libeend_hip_researly.so timeout 300 python -m pytest tests/test_hip_ffn_stream.py -q -x -p no:cacheprovider 2>&1 | tail -1
EEND_HIP_LIB=$V/libeend_hip_spkearly.so timeout 300 python -m pytest tests/test_hip_spk_stream.py -q -x -p no:cacheprovider 2>&1 | tail -1
for r in 1 2; do
echo "== default"; AB_ROUNDS=5 python tools/ab_ffn_stream.py 2>&1 | grep "stream res16"; python tools/ab_spk_stream.py 2>&1 | tail -3
echo "== early"; EEND_HIP_LIB=$V/libeend_hip_researly.so AB_ROUNDS=5 python tools/ab_ffn_stream.py 2>&1 | grep "stream res16"; EEND_HIP_LIB=$V/libeend_hip_spkearly.so python tools/ab_spk_stream.py 2>&1 | tail -3
done
EEND_HIP_LIB=$V/libeend_hip_spkearly.so timeout 200 python tools/ls_breakdown.py 2>&1 | grep -E "frames/s|spk_stream_res32"
timeout 200 python tools/ls_breakdown.py 2>&1 | grep -E "frames/s|spk_stream_res32"
