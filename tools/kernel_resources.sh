#!/bin/bash
# usage: tools/kernel_resources.sh <file.hip> [name filter]   -- registers / spills / occupancy of every kernel in one source
# file, compiled with the flags easynlp_amd/csrc/build.py uses for it
SRC=$1; FILT=${2:-.}
EXTRA=$(python -c "import sys,os; sys.path.insert(0,'easynlp_amd/csrc'); import build as B; print(' '.join(B.FILE_FLAGS.get(os.path.basename('$SRC'), [])))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 --cuda-device-only -O3 -std=c++17 $EXTRA -Iinclude -Ieasynlp_amd/csrc -c $SRC -o /tmp/kr_dev.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|VGPRs Spill|Occupancy|ScratchSize" |
  sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | paste - - - - - | grep -E "$FILT" | c++filt | cut -c1-260
