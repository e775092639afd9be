#!/bin/bash
# usage: tools/isa_stats.sh file.hip  -> per-kernel VGPR / AGPR / scratch / LDS from the gfx950 ISA metadata
set -e
src=$(readlink -f "$1"); tmp=$(mktemp -d); cd "$tmp"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -c "$src" -o x.o -save-temps=obj ${@:2}
python3 - <<'PY'
import re,glob
s=open(glob.glob('*gfx950.s')[0]).read()
for m in re.finditer(r'- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)', s, re.S):
    a,l,n,p,v,sp=m.groups()
    print(f"vgpr {v:>4} agpr {a:>4} scratch {p:>6} spill {sp:>4} lds {l:>6}  {n[:110]}")
PY
rm -rf "$tmp"
