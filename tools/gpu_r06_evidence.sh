#!/bin/bash
# round 6: the evidence set from ONE binary: GPU tests, smoke, bench (all modes), rocprofv3 kernel stats per mode, PMC traffic passes
# (FS and LS inference), the ret_stream phase trace, the 2-process rehearsal.  TAG = suffix of the files (final / mid).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
TAG=${1:-final}; R=$PWD; O=gpurun_out
WHAT="${2:-test smoke bench prof pmc rehearse}"
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has test; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/r06_pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" >> $O/r06_pytest_gpu_$TAG.log; tail -4 $O/r06_pytest_gpu_$TAG.log
fi
if has smoke; then timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> $O/r06_smoke_$TAG.log; tail -2 $O/r06_smoke_$TAG.log; fi
if has bench; then
  timeout 1200 python bench.py > $O/r06_fs_bench_$TAG.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/r06_fs_bench_$TAG.json
  timeout 600 python bench.py --mode train --steps 20 --warmup 3 > $O/r06_train_fs_bench_$TAG.json 2>> $O/bench.err; echo "train rc=$?"; cut -c1-200 $O/r06_train_fs_bench_$TAG.json
  timeout 600 python bench.py --mode train --flavour ls --steps 10 --warmup 3 > $O/r06_train_ls_bench_$TAG.json 2>> $O/bench.err; echo "train ls rc=$?"; cut -c1-200 $O/r06_train_ls_bench_$TAG.json
fi
prof() {   # prof <tag> <command...>
  local tag=$1; shift
  rm -rf $O/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$O/prof" -o $tag -- "$@") > $O/prof_$tag.log 2>&1; echo "prof $tag rc=$?"
  db=$(find $O/prof -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py "$db" $O/r06_${tag}_kernel_stats_$TAG.csv && head -8 $O/r06_${tag}_kernel_stats_$TAG.csv | cut -c1-160
  rm -rf $O/prof
}
if has prof; then
  prof fs python "$R/bench.py" --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-breakdown --graph 0
  prof ls python "$R/tools/ls_prof.py" 5
  prof train_fs python "$R/bench.py" --mode train --steps 3 --warmup 2 --no-breakdown --no-cpu-baseline
  prof train_ls python "$R/bench.py" --mode train --flavour ls --steps 3 --warmup 2 --no-breakdown --no-cpu-baseline
fi
if has pmc; then
  rm -rf $O/pmc
  PMC_PASSES=fetch,write,sq bash tools/gpu_pmc.sh > $O/pmc_fs.log 2>&1
  python tools/pmc_traffic.py $O/pmc/fetch.csv $O/pmc/write.csv $O/r06_pmc_traffic.json
  cp $O/pmc/sq.csv $O/r06_pmc_sq.csv; cp $O/pmc/fetch.csv $O/r06_pmc_fetch.csv; cp $O/pmc/write.csv $O/r06_pmc_write.csv
  PMC_TARGET=ls PMC_PASSES=fetch,write,sq bash tools/gpu_pmc.sh > $O/pmc_ls.log 2>&1
  python tools/pmc_traffic.py $O/pmc/ls_fetch.csv $O/pmc/ls_write.csv $O/r06_ls_pmc_traffic.json "tools/ls_prof.py 2: LS-EEND model.test 16 x T=2000, C=10, eager"
  cp $O/pmc/ls_sq.csv $O/r06_pmc_ls_sq.csv; cp $O/pmc/ls_fetch.csv $O/r06_pmc_ls_fetch.csv; cp $O/pmc/ls_write.csv $O/r06_pmc_ls_write.csv
  bash tools/gpu_r06_pmc_train.sh > $O/pmc_train.log 2>&1
  for fl in fs ls; do for k in sq fetch write; do cp $O/pmc/r06_train_${fl}_$k.csv $O/r06_pmc_train_${fl}_$k.csv 2>/dev/null; done; done
  python - <<'PY'
import json
for f in ("gpurun_out/r06_pmc_traffic.json", "gpurun_out/r06_ls_pmc_traffic.json", "gpurun_out/r06_train_fs_pmc_traffic.json", "gpurun_out/r06_train_ls_pmc_traffic.json"):
    d = json.load(open(f))["kernels"]
    print(f)
    for k, v in list(d.items())[:9]: print("  %8.1f MB  %s" % (v["hbm_bytes"] / 1e6, k[:110]))
PY
fi
if has trace; then
  EEND_HIP_LIB=$R/fs-eend_amd/csrc/variants/libeend_hip_rstrace.so timeout 200 python tools/ret_stream_trace.py > $O/r06_trace_ret_stream_$TAG.txt 2>&1; head -3 $O/r06_trace_ret_stream_$TAG.txt
fi
if has rehearse; then bash tools/gpu_round.sh rehearse > $O/r06_rehearse_$TAG.log 2>&1; tail -6 $O/r06_rehearse_$TAG.log
  for t in infer train_fs train_ls; do cp $O/rehearse_$t.json $O/r06_rehearse_$t.json 2>/dev/null; done; cp $O/rehearse_check.txt $O/r06_rehearse_check.txt 2>/dev/null
fi
