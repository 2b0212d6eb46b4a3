#!/bin/bash
# product-library variants for a same-box A/B (tools/run_ab.sh): tools/build_ab.sh name "-DFLAG=1 ..." [file.hip]   -> easy_vitpose_amd/_lib/ab/name.so
# (only the named translation unit -- default gemm8.hip -- is recompiled with the extra flags; the other objects are the product build's)
set -e
cd $(dirname $0)/..
NAME=$1; DEFS=$2; SRC=${3:-gemm8.hip}
L=easy_vitpose_amd/_lib; mkdir -p $L/ab
python -c "from easy_vitpose_amd.build import build_library; build_library()"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=fast -Wno-unused-result $DEFS \
  -Rpass-analysis=kernel-resource-usage -c easy_vitpose_amd/csrc/$SRC -o $L/ab/$NAME.o 2> $L/ab/$NAME.log
echo -n "$NAME VGPRs / spills: "; grep -E "VGPRs:|VGPRs Spill" $L/ab/$NAME.log | awk '{printf "%s ", $(NF-1)}'; echo
OBJS=""; for s in $(python -c "from easy_vitpose_amd.build import SOURCES; print(' '.join(x[:-4] for x in SOURCES))"); do
  if [ $s.hip = $SRC ]; then OBJS="$OBJS $L/ab/$NAME.o"; else OBJS="$OBJS $L/$s.o"; fi; done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $L/ab/$NAME.so
