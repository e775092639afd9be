cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_hip_kernels.py -q -x -p no:cacheprovider -k head 2>&1 | tail -3
timeout 900 python -m pytest tests/test_fs_parity.py tests/test_ls_parity.py -q -x -p no:cacheprovider 2>&1 | tail -3
R=$PWD; O=gpurun_out; rm -rf $O/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d "$R/$O/prof" -o fs -- python "$R/bench.py" --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-breakdown --graph 0) > $O/prof_fs.log 2>&1
db=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $db $O/tmp_fs_stats.csv >/dev/null; grep -i "head" $O/tmp_fs_stats.csv | cut -c1-200
rm -rf $O/prof
timeout 300 python tools/ls_prof.py 10 2>&1 | tail -1
