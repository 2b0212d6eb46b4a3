mkdir -p gpurun_out/r6i
for f in 8 6 4 2 0; do
  echo "== VP_FOLD_STATS=$f"
  VP_FOLD_STATS=$f VP_HIP_LIB=$PWD/easy_vitpose_amd/_lib/libvitpose_hip.so timeout 400 python tools/small_sweep.py --iters 80 --sets 'default=' --cases l:coco_25:8,l:coco_25:7,l:coco_25:6,l:coco_25:5,l:coco_25:4,l:coco_25:3,l:coco_25:2,l:coco_25:1,b:coco:8,b:coco:6,b:coco:4,b:coco:2,b:coco:1,h:wholebody:8,h:wholebody:6,h:wholebody:4,h:wholebody:2,h:wholebody:1,s:coco:8,s:coco:4,s:coco:1 2>&1 | grep -v amdgpu | cut -c1-100
done > gpurun_out/r6i/fold.txt 2>&1
python - <<'PY'
import re
rows={}; f=None; case=None
for l in open('gpurun_out/r6i/fold.txt'):
    m=re.match(r'== VP_FOLD_STATS=(\d+)',l)
    if m: f=int(m.group(1)); continue
    m=re.match(r'# ViTPose-(\w) / (\w+), (\d+) crops',l)
    if m: case=(m.group(1),int(m.group(3))); continue
    m=re.match(r'default\s+([\d.]+) ms',l)
    if m: rows.setdefault(case,{})[f]=float(m.group(1))
for c,v in rows.items():
    print(c, ' '.join(f'fold<={k}: {v[k]:.3f}' for k in sorted(v)))
PY
