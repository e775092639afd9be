#!/bin/bash
# round 6, call D: training-step kernel statistics (FS, LS) with the new weight-gradient kernel; sustained-launch behaviour of the kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD
for fl in fs ls; do
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o tr -- python "$R/bench.py" --mode train --flavour $fl --steps 5 --warmup 2 --no-cpu-baseline --no-breakdown) > gpurun_out/prof_train_$fl.log 2>&1
  echo "prof $fl rc=$?"; tail -1 gpurun_out/prof_train_$fl.log | cut -c1-300
  db=$(find gpurun_out/prof -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/r06_train_${fl}_kernel_stats_mid.csv && head -40 gpurun_out/r06_train_${fl}_kernel_stats_mid.csv | cut -c1-200
  rm -rf gpurun_out/prof
done
# 60 back-to-back launches: do the later ones run slower than the first four?
cat > /tmp/many.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["R"])
from fs_eend_amd.train import _call, WS_FLOATS
M, N, K = 196608, 2048, 256
dev = torch.device("cuda")
ws = torch.empty(WS_FLOATS, dtype=torch.float32, device=dev)
dy = (torch.randn(M, N, device=dev) * 1e-3).to(torch.bfloat16)
x = torch.randn(M, K, device=dev).to(torch.float16)
out = torch.empty(N, K, dtype=torch.float32, device=dev)
torch.cuda.synchronize()
for _ in range(60):
    _call("eend_wgrad_bf16", dy, N, x, K, 1, M, N, K, ws, WS_FLOATS, out, K, K, 1.0, 0)
torch.cuda.synchronize()
PY
rm -rf /tmp/tr_many
(cd /tmp && R=$R timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_many -o t -- python /tmp/many.py) > /tmp/tr_many.log 2>&1
f=$(find /tmp/tr_many -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "wgrad_tr_kernel" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("wgrad_tr durations over 60 back-to-back launches (us):", " ".join(f"{x:.0f}" for x in d))
PY
