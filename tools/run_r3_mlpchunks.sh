#!/bin/bash
# fc1 -> fc2 over row chunks (VP_MLP_CHUNKS): bit-equality with the whole-batch MLP and same-box A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/mlpchunks.txt; rm -f $O
for n in 1 3; do
VP_MLP_CHUNKS=$n timeout 300 python - >> $O 2>&1 <<PY
import numpy as np, hashlib, os
from easy_vitpose_amd.engine import VitPoseHip
from easy_vitpose_amd.configs import model_shape
from easy_vitpose_amd.synth import synthetic_state_dict
shp = model_shape('b', 'coco')
eng = VitPoseHip(shp, synthetic_state_dict(shp, seed=3, peaked=True), dtype='fp16', max_batch=256)
crops = np.random.default_rng(5).standard_normal((256, 3, 256, 192)).astype(np.float32)
kp = np.asarray(eng.infer(crops))
print('chunks', os.environ['VP_MLP_CHUNKS'], 'keypoints sha', hashlib.sha256(np.ascontiguousarray(kp).tobytes()).hexdigest()[:16], kp.shape, float(np.abs(kp).max()))
eng.close()
PY
done
for r in 1 2 3; do for n in 1 3 2 4; do
  echo -n "chunks $n: " >> $O
  VP_MLP_CHUNKS=$n timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-path --no-clock --breakdown 2>&1 | python -c "
import sys,json
o=''
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        if 'breakdown' in d: o=' '.join(f'{k[5:9] if k.startswith(\"gemm\") else k[:5]}={v[\"ms_per_step\"]:.3f}' for k,v in d['breakdown'].items())
        else: print(d['value'], d['ms_per_step'], o)
" >> $O
done; done
cat $O
