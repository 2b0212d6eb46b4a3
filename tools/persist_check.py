#!/usr/bin/env python3
"""Race screen for the persistent GEMM variant: heatmaps with VP_PERSIST=1 must be bit-identical to VP_PERSIST=0
(same arithmetic, different tile schedule), over repeated runs and several batch sizes.  GPU box only."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _toolslib  # noqa: F401,E402  (measurement build of the library)

if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import hashlib
    import numpy as np
    from easy_vitpose_amd.configs import model_shape
    from easy_vitpose_amd.engine import VitPoseHip
    from easy_vitpose_amd.synth import synthetic_crops, synthetic_state_dict
    variant, batch, reps = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    shp = model_shape(variant, 'coco')
    eng = VitPoseHip(shp, synthetic_state_dict(shp, 0), dtype='fp16', max_batch=batch)
    crops = synthetic_crops(batch, 3, 'noise')
    for r in range(reps):
        out = eng.infer(crops)
        print(hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest()[:16], flush=True)
    eng.close()
    sys.exit(0)

ok = True
for variant, batch in (('b', 64), ('b', 256), ('s', 128), ('l', 96), ('b', 57), ('b', 100), ('h', 43)):
    hashes = {}
    for flag in ('0', '1'):
        env = dict(os.environ, VP_PERSIST=flag)
        res = subprocess.run([sys.executable, __file__, 'child', variant, str(batch), '6'], env=env, capture_output=True, text=True)
        hashes[flag] = [l for l in res.stdout.split() if len(l) == 16]
        if res.returncode != 0:
            print(res.stderr[-2000:])
    allh = sum(hashes.values(), [])
    same = len(set(allh)) == 1 and all(len(v) == 6 for v in hashes.values())
    ok &= same
    print(f'ViTPose-{variant.upper()} batch {batch}: ' + ' '.join(f'persist={k} {sorted(set(v))}' for k, v in hashes.items()) + f' -> {"IDENTICAL" if same else "MISMATCH"}')
sys.exit(0 if ok else 1)
