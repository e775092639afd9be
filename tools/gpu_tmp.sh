cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
run() { timeout 600 python -c "
import sys, runpy
sys.path.insert(0, '.')
import fs_eend_amd
from fs_eend_amd import train_ls as t
t.LsTrainStep.proj_stream_min_rows = $1
sys.argv = ['bench.py', '--mode', 'train', '--flavour', 'ls', '--steps', '10', '--warmup', '3', '--no-breakdown', '--no-cpu-baseline']
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3))"; }
for i in 1 2 3; do run 1000000000; run 49152; done
