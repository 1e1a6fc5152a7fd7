#!/bin/bash
# usage: tools/kres.sh <kernel-name-substring> : VGPRs / scratch / occupancy and the instruction count of a kernel of cc_engine.hip (gfx950 cross-compile)
cd /root/repo/continuous_clustering_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -S --cuda-device-only -o /tmp/cc_engine.s cc_engine.hip -Rpass-analysis=kernel-resource-usage 2> /tmp/cc_engine.res
grep -E "error" -A5 /tmp/cc_engine.res | head -30
for k in "$@"; do
  grep -A7 "Function Name: _ZN3cck[0-9]*$k" /tmp/cc_engine.res | grep -E "Function Name|VGPRs:|Scratch|Occupancy|LDS" | sed 's/.*remark: *//'
  for n in $(grep -n "^_ZN3cck[0-9]*$k.*:" /tmp/cc_engine.s | cut -d: -f1); do
    awk -v n=$n 'NR>=n{print} /^\.Lfunc_end/{if(NR>n){exit}}' /tmp/cc_engine.s > /tmp/kres_$k.s
    echo "asm lines: $(wc -l < /tmp/kres_$k.s) scratch ops: $(grep -c scratch_ /tmp/kres_$k.s)"
  done
done
