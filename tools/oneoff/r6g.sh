mkdir -p gpurun_out/r6g
(timeout 600 python -m pytest tests/test_gpu_production_sizes.py -k "content" -q -s -p no:cacheprovider 2>&1 | grep "^\[content\|^    crop\|passed\|failed" > gpurun_out/r6g/content_lines.txt)
for ab in "" "1:1" "1:8" "1:9" "1:2" "2:1,1:1,0:1,10:1" "2:9,1:9,0:9,10:9"; do
  echo "== VP_ABLATE_FAM=$ab (fam 1 = fc1, 2 = qkv, 0 = proj, 10 = fc2; bits: 1 = no operand loads after the prologue, 2 = every tile loads the A rows of m-tile 0, 8 = no epilogue stores)"
  VP_ABLATE_FAM=$ab timeout 200 python tools/small_sweep.py --iters 60 --cases l:coco_25:8 --sets 'default=' 2>&1 | grep -v amdgpu | cut -c1-160
done > gpurun_out/r6g/ablate_l8.txt 2>&1
cat gpurun_out/r6g/content_lines.txt | cut -c1-500; cat gpurun_out/r6g/ablate_l8.txt
