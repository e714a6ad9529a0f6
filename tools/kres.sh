#!/bin/bash
# tools/kres.sh FILE.hip [grep-pattern]: per-kernel registers / spills / occupancy (hipcc -Rpass-analysis=kernel-resource-usage)
f=$1; pat=${2:-.}
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I /root/repo/include -I /root/repo/gpt4roi_amd/csrc -c "$f" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep remark | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | awk '/Function Name/{if(l)print l; l=$0; next}{l=l" | "$0}END{print l}' \
 | while read -r line; do name=$(echo "$line" | sed -e 's/Function Name: \([^ ]*\).*/\1/' | c++filt | cut -c1-110); echo "$name :: $(echo "$line" | grep -oE '(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): [0-9]+' | tr '\n' ' ')"; done | grep -E "$pat"
