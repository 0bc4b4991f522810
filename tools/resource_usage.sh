#!/bin/bash
# Prints VGPR/AGPR/SGPR/spill/occupancy per kernel of one .hip file (compiler view).
f=$1
cd "$(dirname "$f")"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c "$(basename "$f")" -o /tmp/_ru.o 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs:|SGPRs Spill|VGPRs Spill|Occupancy|ScratchSize" \
 | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
 | awk '/Function Name/{if(l)print l; l=$0; next}{l=l" | "$0}END{print l}'
