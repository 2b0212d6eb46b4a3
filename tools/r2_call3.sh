#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c3; O=gpurun_out/c3
timeout 300 python tools/determinism_check.py s 16 6 > $O/det_s16.txt 2>&1
VP_FUSE_LN=0 timeout 300 python tools/determinism_check.py s 16 4 > $O/det_s16_nofuse.txt 2>&1
timeout 300 python tools/determinism_check.py b 8 4 > $O/det_b8.txt 2>&1
timeout 300 python tools/determinism_check.py s 4 4 > $O/det_s4.txt 2>&1
cat $O/det_s16.txt $O/det_s16_nofuse.txt $O/det_b8.txt $O/det_s4.txt
