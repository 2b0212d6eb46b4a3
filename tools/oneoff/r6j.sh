S='default=;w33=QKV:33:0,FC1:33:0;w34=QKV:34:0,FC1:34:0;w35=QKV:35:0,FC1:35:0;w33g=QKV:33:8,FC1:33:8;w34g=QKV:34:8,FC1:34:8;r34=QKV:34:0,FC1:34:0,PROJ:34:0,FC2:34:0;default_b='
VP_FOLD_STATS=0 timeout 900 python tools/small_sweep.py --iters 60 --cases l:coco_25:8,l:coco_25:6,l:coco_25:4,h:wholebody:8,b:coco:8,l:coco_25:12,l:coco_25:16,b:coco:16 --sets "$S" 2>&1 | grep -v amdgpu | cut -c1-250
