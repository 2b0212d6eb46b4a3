#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
AB=$PWD/easy_vitpose_amd/_lib/ab
VP_HIP_LIB=$AB/tools_abl16.so timeout 300 python tools/gemm8_timeline.py > gpurun_out/r3_sched2_sections.txt 2>&1; grep -v amdgpu gpurun_out/r3_sched2_sections.txt
timeout 300 python tools/gemm8_timeline.py > gpurun_out/r3_sched2_timeline.txt 2>&1; grep -v amdgpu gpurun_out/r3_sched2_timeline.txt | head -30
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3_pytest_sched2.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3_pytest_sched2.log | head -2
