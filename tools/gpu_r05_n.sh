#!/bin/bash
# round 5 extras for the record: layer-head kernel phase traces (FS form and the LS f32-residual form), FS kernel stats under graph replay
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1; R=$PWD; O=gpurun_out
V=$R/fs-eend_amd/csrc/variants
EEND_HIP_LIB=$V/libeend_hip_spktrace.so timeout 200 python tools/spk_stream_trace.py > $O/r05_trace_spk_stream_fs.txt 2>&1; tail -14 $O/r05_trace_spk_stream_fs.txt
SPK_TRACE_R32=1 EEND_HIP_LIB=$V/libeend_hip_spktrace.so timeout 200 python tools/spk_stream_trace.py > $O/r05_trace_spk_stream_r32.txt 2>&1; tail -14 $O/r05_trace_spk_stream_r32.txt
rm -rf $O/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$O/prof" -o fsg -- python "$R/bench.py" --steps 20 --warmup 2 --no-cpu-baseline --no-extras --no-breakdown --graph 1) > $O/prof_fsg.log 2>&1; echo "prof rc=$?"
db=$(find $O/prof -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" $O/r05_fs_kernel_stats_graph_final.csv && head -12 $O/r05_fs_kernel_stats_graph_final.csv | cut -c1-160
python - <<PY
import sqlite3
db = sqlite3.connect("$db")
rows = db.execute("select name, start, end from kernels order by start").fetchall()
# gaps between consecutive kernels inside the steady replays (last 300 kernels)
tail = rows[-360:]
gaps = [(b[1] - a[2]) for a, b in zip(tail[:-1], tail[1:]) if 0 <= b[1] - a[2] < 200000]
busy = sum(r[2] - r[1] for r in tail)
span = tail[-1][2] - tail[0][1]
print("last 360 kernels: span %.1f us, kernel time %.1f us (%.1f %%), mean gap %.2f us, median %.2f us" % (span / 1e3, busy / 1e3, 100.0 * busy / span, sum(gaps) / len(gaps) / 1e3, sorted(gaps)[len(gaps) // 2] / 1e3))
PY
rm -rf $O/prof
