#!/bin/bash
# round 6, call C: pipelined weight-gradient loop: tests, kernel trace per variant / shape, step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
ROOT=$PWD; V=$ROOT/fs-eend_amd/csrc/variants
timeout 900 python -m pytest tests/test_train_kernels.py -q -x -k "wgrad or conv1d" -p no:cacheprovider 2>&1 | tail -3
trace() {  # tag, lib, args of ab_wgrad_one.py
  tag=$1; lib=$2; shift 2
  rm -rf /tmp/tr_$tag
  (cd /tmp && EEND_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$tag -o t -- python $ROOT/tools/ab_wgrad_one.py "$@") > /tmp/tr_$tag.log 2>&1
  f=$(find /tmp/tr_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag [$*] X_F16=${X_F16:-1}"; python tools/kstats.py $f wgrad
}
NEW=$ROOT/fs-eend_amd/csrc/libeend_hip.so
for shape in "196608 2048 256" "196608 256 2048" "196608 256 256" "32768 256 256" "32768 2048 256" "393216 2048 256" "196608 768 256"; do
  trace new $NEW $shape
done
X_F16=0 trace new $NEW 196608 2048 256
X_F16=0 trace new $NEW 196608 256 256
for v in v1 nodma noread; do trace $v $V/libeend_hip_$v.so 196608 2048 256; done
for v in v1 f128 nodma; do trace $v $V/libeend_hip_$v.so 196608 256 256; done
for v in v1 f128; do trace $v $V/libeend_hip_$v.so 32768 256 256; done
for v in v1 f128; do trace $v $V/libeend_hip_$v.so 196608 768 256; done
for r in 1 2; do for lib in new v1 r05; do
  if [ $lib = new ]; then unset EEND_HIP_LIB; else export EEND_HIP_LIB=$V/libeend_hip_$lib.so; fi
  timeout 300 python bench.py --mode train --steps 20 --warmup 3 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib FS', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --mode train --flavour ls --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib LS', d['value'], d['ms_per_step'])"
done; done
unset EEND_HIP_LIB
