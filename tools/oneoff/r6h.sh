mkdir -p gpurun_out/r6h
S='default=;w18=QKV:18:4,FC1:18:8;w16=QKV:16:4,FC1:16:8;f18=FC1:18:8;q18=QKV:18:4;w18r=QKV:18:4,FC1:18:8,FC2:18:2,PROJ:18:2;f2_18=FC2:18:2;default_b='
echo "== VP_FOLD_STATS default (8)"; timeout 300 python tools/small_sweep.py --iters 60 --cases l:coco_25:8 --sets 'default=' 2>&1 | grep -v amdgpu | cut -c1-230
echo "== VP_FOLD_STATS=0 (ln_finalize launches; the 8-phase kernel cannot merge partial statistics itself)"
VP_FOLD_STATS=0 timeout 500 python tools/small_sweep.py --iters 60 --cases l:coco_25:8,h:wholebody:8,b:coco:8,l:coco_25:4,l:coco_25:12 --sets "$S" 2>&1 | grep -v amdgpu | cut -c1-230
