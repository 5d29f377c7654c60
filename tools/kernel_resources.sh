#!/bin/bash
# usage: tools/kernel_resources.sh <file.hip> [name filter]   -- registers / spills / occupancy of every kernel in one source file
SRC=$1; FILT=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 --cuda-device-only -O3 -std=c++17 -Iinclude -Ieasynlp_amd/csrc -c $SRC -o /tmp/kr_dev.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|VGPRs Spill|Occupancy|ScratchSize" |
  sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | paste - - - - - | grep -E "$FILT" | c++filt | cut -c1-260
