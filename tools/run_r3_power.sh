#!/bin/bash
# shader clock + board power while each kernel runs; the 8-phase kernel with pieces removed (run-time ablations and -DVP_G8_ABL builds)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/clock_power.txt; rm -f $O
timeout 400 python tools/clock_power_probe.py --secs 5 >> $O 2>&1
for a in 1 4 5; do
  echo "== tools build with -DVP_G8_ABL=$a (1 = no fragment reads after a tile's first K-tile, 4 = no MFMAs)" >> $O
  VP_PROBE_SET=abl VP_HIP_LIB=$PWD/easy_vitpose_amd/_lib/ab/abl$a.so timeout 200 python tools/clock_power_probe.py --secs 5 2>&1 | grep "fc1 shape" >> $O
done
cat $O
