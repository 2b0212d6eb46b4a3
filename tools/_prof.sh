cd $GRAFT_REPO_ROOT
bash tools/profile.sh r2 > gpurun_out/prof_r2.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_r2 gpurun_out/pmc_r2.json > gpurun_out/r2_rocprofv3.txt 2>&1
# drop the big raw databases, keep the summaries
find gpurun_out/prof_r2 -name "*.db" -size +8M -delete
tail -5 gpurun_out/r2_rocprofv3.txt; head -22 gpurun_out/r2_rocprofv3.txt
