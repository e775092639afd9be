cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_fs_parity.py tests/test_ls_parity.py -m gpu -q -x --timeout 300 -p no:cacheprovider -k "ffn or attnout or golden or parity or fused" 2>&1 | tail -3
for r in 1 2; do echo "== default (SRC=2)"; python tools/ab_ops.py 2>&1 | grep -E "attnout|linear_res_ln"; echo "== presrc0"; EEND_HIP_LIB=$PWD/fs-eend_amd/csrc/variants/libeend_hip_presrc0.so python tools/ab_ops.py 2>&1 | grep -E "attnout|linear_res_ln"; done
EEND_HIP_LIB=$PWD/fs-eend_amd/csrc/variants/libeend_hip_ffntrace.so timeout 300 python tools/ffn_trace.py 2>&1 | tail -9
