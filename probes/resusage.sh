#!/bin/bash
# compact per-kernel resource usage: probes/resusage.sh file.hip
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -c "$1" -o /tmp/_ru.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|VGPRs Spill|ScratchSize|Occupancy|LDS Size" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - - - 
