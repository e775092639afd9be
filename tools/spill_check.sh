#!/bin/bash
# tools/spill_check.sh <source stem> "<-Dflags>" <kernel mangled-name substring>: register report + scratch map of one kernel
CS=/root/repo/fs-eend_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $2 -S --cuda-device-only $CS/$1.hip -o /tmp/$1_chk.s -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs Spill" | sed 's/.*remark: [^ ]* *//; s/\[-Rpass.*//' | paste - - | grep "error\|$3"
name=$(grep -o "^_Z[A-Za-z0-9_]*$3[A-Za-z0-9_]*:" /tmp/$1_chk.s | head -1 | tr -d ':')
python /root/repo/tools/spill_map.py /tmp/$1_chk.s $name | fold -w 220
