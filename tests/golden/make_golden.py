#!/usr/bin/env python3
"""Generate the golden fixtures by running THE REFERENCE ITSELF in this container.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz
    python tests/golden/make_golden.py --only=tracker,outlier     # sections: caller decode model tracker peaked outlier fullbatch content

Runs only where ``/root/reference`` exists (the build container).  The reference
package is imported read-only with ``sys.dont_write_bytecode`` and with empty stub
modules for the six third-party imports that are not installed here
(cv2, ultralytics, skimage, filterpy, torchvision, ffmpeg -- SURVEY.md 8c).  Two
cv2 functions are on the path and get a real implementation in the stub:

* ``cv2.GaussianBlur`` -> scipy.ndimage.correlate1d(mode='mirror') on both axes with
  OpenCV's float32 ``getGaussianKernel`` weights (an implementation independent of
  ``oracle/vitpose_cpu.gaussian_blur``; PARITY UNPINNED vs the real OpenCV binary),
* ``cv2.resize`` -> identity when the crop is already 256x192; for the caller-level golden
  (``frame_inference.npz``, crops of arbitrary size) the ORACLE's plain-C restatement of OpenCV's 8-bit
  INTER_LINEAR (``oracle/resize_ref.c`` through ``oracle.resize_ref.resize_linear_u8``; no golden depends on
  product code -- the product's ``cropprep.resize_linear_u8`` and the device kernel are TESTED against it) -- the resize boundary stays
  PARITY UNPINNED, the golden pins everything around it (box padding/clipping, ``pad_image``,
  the per-box loop, the offset arithmetic, the key -> id mapping).

Nothing from the reference is copied: the fixtures hold only seeds/shapes and the
reference's numerical OUTPUTS.  While generating, the script also checks the
``oracle/`` restatement against the reference on identical inputs and prints the
deviations (the same checks that tests/test_oracle_golden.py repeats against the
stored outputs).
"""
from __future__ import annotations

import os
import sys
import types
import warnings

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np
import torch

REF = '/root/reference'


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Stub(f'{self.__name__}.{name}')

    def __call__(self, *a, **k):
        raise RuntimeError(f'stub {self.__name__} called')


def _install_stubs():
    from scipy.ndimage import correlate1d

    for name in ['cv2', 'ultralytics', 'skimage', 'skimage.io', 'filterpy', 'filterpy.kalman',
                 'torchvision', 'torchvision.transforms', 'torchvision.utils', 'ffmpeg',
                 'matplotlib', 'matplotlib.pyplot', 'matplotlib.patches', 'matplotlib.cm']:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = _Stub(name)
    cv2 = sys.modules['cv2']

    def getGaussianKernel(ksize, sigma):
        if sigma <= 0:
            sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
        i = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
        w = np.exp(-(i * i) / (2 * sigma * sigma))
        return (w / w.sum()).astype(np.float32)

    def GaussianBlur(src, ksize, sigmaX, dst=None, *a, **k):
        g = getGaussianKernel(ksize[0], sigmaX)
        out = correlate1d(src.astype(np.float32), g, axis=1, mode='mirror')
        out = correlate1d(out, g, axis=0, mode='mirror').astype(np.float32)
        if dst is not None:
            dst[...] = out
            return dst
        return out

    def resize(img, dsize, interpolation=None):
        if (img.shape[1], img.shape[0]) == tuple(dsize):
            return img
        from oracle.resize_ref import resize_linear_u8     # the checker's C restatement, never the product's resize
        assert img.dtype == np.uint8
        return resize_linear_u8(np.ascontiguousarray(img), list(dsize))

    cv2.GaussianBlur = GaussianBlur
    cv2.resize = resize
    cv2.INTER_LINEAR = 1

    # filterpy (absent) -- the reference's tracker (sort.py:30,104-119) only uses KalmanFilter(dim_x, dim_z) with the attributes
    # x, P, Q, R, F, H and predict() / update(z): the textbook linear Kalman filter, restated here from filterpy's documented
    # equations (P update in Joseph form).  PARITY UNPINNED vs the filterpy package itself.
    class KalmanFilter:
        def __init__(self, dim_x, dim_z):
            self.x = np.zeros((dim_x, 1)); self.P = np.eye(dim_x); self.Q = np.eye(dim_x)
            self.R = np.eye(dim_z); self.F = np.eye(dim_x); self.H = np.zeros((dim_z, dim_x))
            self._I = np.eye(dim_x)

        def predict(self):
            self.x = self.F @ self.x
            self.P = self.F @ self.P @ self.F.T + self.Q

        def update(self, z):
            z = np.asarray(z, dtype=np.float64).reshape(-1, 1)
            y = z - self.H @ self.x
            S = self.H @ self.P @ self.H.T + self.R
            K = self.P @ self.H.T @ np.linalg.inv(S)
            self.x = self.x + K @ y
            IKH = self._I - K @ self.H
            self.P = IKH @ self.P @ IKH.T + K @ self.R @ K.T
    sys.modules['filterpy.kalman'].KalmanFilter = KalmanFilter


def import_reference():
    _install_stubs()
    sys.path.insert(0, REF)
    warnings.simplefilter('ignore')
    import easy_ViTPose  # noqa: F401
    from easy_ViTPose.inference import VitInference
    from easy_ViTPose.vit_models.model import ViTPose
    from easy_ViTPose.vit_utils.util import dyn_model_import
    return VitInference, ViTPose, dyn_model_import


def build_ref(VitInference, ViTPose, dyn_model_import, dataset, variant, sd_np):
    cfg = dyn_model_import(dataset, variant)
    # the reference's config modules share ONE dict per variant and set out_channels at import time only: a second dataset's
    # model built later would keep the first one's K.  Set what ViTPose_<dataset>.py sets.
    cfg['keypoint_head']['out_channels'] = int(sd_np['keypoint_head.final_layer.bias'].shape[0])
    model = ViTPose(cfg)
    model.eval()
    missing = model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    assert not missing.missing_keys and not missing.unexpected_keys, missing
    V = VitInference.__new__(VitInference)
    V.device = 'cpu'
    V.target_size = [192, 256]
    V._vit_pose = model
    return V


def main():
    from cases import org_sizes, peaked_heatmaps
    from easy_vitpose_amd.configs import model_shape
    from easy_vitpose_amd.synth import synthetic_crops, synthetic_state_dict
    from oracle import vitpose_cpu as O

    VitInference, ViTPose, dyn_model_import = import_reference()
    torch.manual_seed(0)
    only = [a.split('=', 1)[1].split(',') for a in sys.argv[1:] if a.startswith('--only=')]
    only = set(only[0]) if only else None

    def want(section):   # python tests/golden/make_golden.py --only=tracker,outlier regenerates just those fixtures
        return only is None or section in only

    # ---------------------------------------------------------------- caller goldens
    if want('caller'):
        # (first: the reference's config modules share one dict, a 'coco' model built after 'wholebody' keeps K = 133)
        # (1) the reference's pad_image on seeded crops of assorted shapes
        from easy_ViTPose.vit_utils.inference import pad_image as ref_pad_image
        from cases import frame_case, pad_shapes
        rows = []
        for i, (h, w) in enumerate(pad_shapes()):
            img = np.random.default_rng(100 + i).integers(0, 256, size=(h, w, 3), dtype=np.uint8)
            out, (left, top) = ref_pad_image(img, 3 / 4)
            rows.append([h, w, out.shape[0], out.shape[1], left, top, int(out.astype(np.int64).sum()),
                         int((out[top:top + h, left:left + w] != img).sum())])
        np.savez_compressed(os.path.join(HERE, 'pad_image.npz'), rows=np.asarray(rows, dtype=np.int64))

        # (1b) the reference's flip_back on seeded heatmaps (mirror pairs of COCO-17: 1-2, 3-4, ..., 15-16)
        from easy_ViTPose.vit_utils.post_processing.post_transforms import flip_back as ref_flip_back
        from cases import coco_flip_pairs
        fh = np.random.default_rng(33).standard_normal((2, 17, 64, 48)).astype(np.float32)
        fb = ref_flip_back(fh.copy(), coco_flip_pairs(), target_type='GaussianHeatmap')
        print(f'flip_back: oracle-vs-reference max|d| = {np.abs(O.flip_back(fh, coco_flip_pairs()) - fb).max():.3e}')
        np.savez_compressed(os.path.join(HERE, 'flip_back.npz'), seed=33, expected=np.ascontiguousarray(fb))

        # (2) the reference's VitInference.inference(img) end to end on a synthetic frame: a fake detector object with
        #     the ultralytics result interface feeds the reference's own box loop (inference.py:221-281)
        frame, boxes = frame_case()

        class _Arr:
            def __init__(self, a): self.a = a
            def cpu(self): return self
            def numpy(self): return self.a

        class _Res:
            def __init__(self, a): self.boxes = types.SimpleNamespace(data=_Arr(a))

        shp = model_shape('s', 'coco')
        sd = synthetic_state_dict(shp, seed=0)
        V = build_ref(VitInference, ViTPose, dyn_model_import, 'coco', 's', sd)
        V.yolo = lambda img, **kw: [_Res(boxes.copy())]
        V.tracker = None
        V.frame_counter = 0
        V.yolo_step = 1
        V.yolo_size = 320
        V.yolo_classes = [0]
        V.save_state = True
        V._inference = V._inference_torch
        with torch.no_grad():
            res = V.inference(frame.copy())
        ids = sorted(res.keys())
        kp = np.stack([res[i] for i in ids]).astype(np.float32)
        tb, tids, tscores = V._tracker_res
        print(f'frame golden: {len(ids)} persons, padded boxes {np.asarray(tb).tolist()}')
        np.savez_compressed(os.path.join(HERE, 'frame_inference.npz'), ids=np.asarray(ids), keypoints=kp,
                            padded_boxes=np.asarray(tb, dtype=np.int64), scores=np.asarray(tscores, dtype=np.float64))
    # ---------------------------------------------------------------- decode goldens
    if want('decode'):
        for tag, (n, k, seed) in {'decode_k17': (8, 17, 11), 'decode_k133': (2, 133, 12)}.items():
            hm = peaked_heatmaps(n, k, seed)
            wh = org_sizes(n, seed)
            exp = np.concatenate([VitInference.postprocess(hm[i:i + 1].copy(), int(wh[i, 0]), int(wh[i, 1]))
                                  for i in range(n)], 0).astype(np.float32)
            mine = O.decode_per_crop(hm, wh)
            print(f'{tag}: oracle-vs-reference max|d| = {np.abs(mine - exp).max():.3e}')
            np.savez_compressed(os.path.join(HERE, f'{tag}.npz'), n=n, k=k, seed=seed, org_wh=wh, expected=exp)
    # ----------------------------------------------------------------- model goldens
    if want('model'):
        # (variant, dataset, n_crops, crop kind, channels kept in the fixture)
        plan = [('s', 'coco', 2, 'noise', None), ('b', 'coco', 2, 'blobs', None),
                ('l', 'coco_25', 1, 'blobs', None), ('h', 'wholebody', 1, 'noise', 16)]
        for variant, dataset, n, kind, keep in plan:
            shp = model_shape(variant, dataset)
            sd = synthetic_state_dict(shp, seed=0)
            V = build_ref(VitInference, ViTPose, dyn_model_import, dataset, variant, sd)
            crops = synthetic_crops(n, seed=7, kind=kind)
            hms, kps = [], []
            with torch.no_grad():
                for i in range(n):
                    x, oh, ow = V.pre_img(crops[i])
                    hms.append(V._vit_pose(torch.from_numpy(x)).numpy())
                    kps.append(V._inference_torch(crops[i]))
            hms = np.concatenate(hms, 0)
            kps = np.concatenate(kps, 0).astype(np.float32)
            sdt = O.to_torch_state_dict(sd)
            mine_hm = np.concatenate([O.model_forward(sdt, O.pre_img(crops[i])[0], shp.depth, shp.num_heads)
                                      for i in range(n)], 0)
            mine_kp = np.concatenate([O.inference_torch(sdt, shp.depth, shp.num_heads, crops[i]) for i in range(n)], 0)
            print(f'model {variant}/{dataset}: heatmap max|d| = {np.abs(mine_hm - hms).max():.3e} '
                  f'(hm std {hms.std():.3f}), keypoints max|d| = {np.abs(mine_kp - kps).max():.3e}')
            stats = np.array([hms.mean(), hms.std(), hms.min(), hms.max()], dtype=np.float64)
            hm_store = hms if keep is None else hms[:, :keep]
            np.savez_compressed(os.path.join(HERE, f'model_{variant}_{dataset}.npz'), variant=variant, dataset=dataset,
                                n=n, kind=kind, crop_seed=7, weight_seed=0, heatmaps=hm_store.astype(np.float32),
                                keypoints=kps, stats=stats)
    # ------------------------------------------------------- tracker golden (f-4)
    if want('tracker'):
        # the reference's Sort (sort.py:203-266, constructed as inference.py:182-184 does) on a seeded detection sequence: moving
        # boxes, a missed detection, a late entry, a crossing pair, detector-skipped frames (empty input -> predicted boxes)
        from easy_ViTPose.sort import Sort as RefSort, KalmanBoxTracker
        from cases import tracker_sequence
        for tag, max_age in (('sort_age1', 1), ('sort_age3', 3)):
            KalmanBoxTracker.count = 0
            trk = RefSort(max_age=max_age, min_hits=3, iou_threshold=0.3)
            outs = [np.asarray(trk.update(d.copy()), dtype=np.float64).reshape(-1, 6) for d in tracker_sequence()]
            flat = np.concatenate([np.concatenate([np.full((len(o), 1), i, dtype=np.float64), o], 1) for i, o in enumerate(outs)])
            print(f'tracker golden {tag}: {len(outs)} frames, {len(flat)} reported boxes, ids {sorted(set(flat[:, 6].astype(int)))}')
            np.savez_compressed(os.path.join(HERE, f'{tag}.npz'), rows=flat, max_age=max_age)
        # yolo_step > 1 (detector-skipped frames): the reference builds Sort(max_age=step, min_hits=1) there (inference.py:179-184)
        from cases import tracker_sequence_step
        for step in (2, 3):
            KalmanBoxTracker.count = 0
            trk = RefSort(max_age=step, min_hits=1, iou_threshold=0.3)
            outs = [np.asarray(trk.update(d.copy()), dtype=np.float64).reshape(-1, 6) for d in tracker_sequence_step(step)]
            flat = np.concatenate([np.concatenate([np.full((len(o), 1), i, dtype=np.float64), o], 1) for i, o in enumerate(outs)])
            print(f'tracker golden sort_step{step}: {len(outs)} frames, boxes per frame {[len(o) for o in outs]}')
            np.savez_compressed(os.path.join(HERE, f'sort_step{step}.npz'), rows=flat, max_age=step, min_hits=1)
    # ------------------------------------------------------- peaked-checkpoint goldens
    if want('peaked'):
        # synthetic_state_dict(peaked=True): one Gaussian-like blob per joint, so the reference's own keypoints are
        # well-conditioned on EVERY joint and can be asserted end to end (+-0.5 px, 1e-3) on the device.
        from cases import peaked_plan, peaked_crops
        for variant, dataset, n in peaked_plan():
            shp = model_shape(variant, dataset)
            sd = synthetic_state_dict(shp, seed=0, peaked=True)
            V = build_ref(VitInference, ViTPose, dyn_model_import, dataset, variant, sd)
            crops = peaked_crops(n)
            kps, hm0 = [], None
            with torch.no_grad():
                for i in range(n):
                    kps.append(V._inference_torch(crops[i]))
                    if i == 0:
                        hm0 = V._vit_pose(torch.from_numpy(V.pre_img(crops[0])[0])).numpy()
            kps = np.concatenate(kps, 0).astype(np.float32)
            sdt = O.to_torch_state_dict(sd)
            mine = np.concatenate([O.inference_torch(sdt, shp.depth, shp.num_heads, crops[i]) for i in range(n)], 0)
            print(f'peaked {variant}/{dataset}: {n} crops x {shp.num_keypoints} joints, confidences {kps[..., 2].min():.3f} .. {kps[..., 2].max():.3f}, '
                  f'oracle-vs-reference keypoints max|d| = {np.abs(mine - kps).max():.3e}')
            np.savez_compressed(os.path.join(HERE, f'peaked_{variant}_{dataset}.npz'), variant=variant, dataset=dataset, n=n,
                                keypoints=kps, heatmaps0=hm0[:, :16].astype(np.float32))

    # ------------------------------------------------------- peaked checkpoints WITH activation outliers
    # synthetic_state_dict(peaked=True, outliers=True): the regime of trained ViTs that seeded random tensors lack -- four
    # residual channels at 100-1000 x the scale of the others from block 3 on (two constant, two token-dependent) and an
    # attention head with logits of magnitude ~45.  Goldens = the reference's own keypoints, every joint.
    if want('outlier'):
        from cases import peaked_plan, peaked_crops
        for variant, dataset, n in peaked_plan():
            shp = model_shape(variant, dataset)
            sd = synthetic_state_dict(shp, seed=0, peaked=True, outliers=True)
            V = build_ref(VitInference, ViTPose, dyn_model_import, dataset, variant, sd)
            crops = peaked_crops(n)
            kps, hm0 = [], None
            with torch.no_grad():
                for i in range(n):
                    kps.append(V._inference_torch(crops[i]))
                    if i == 0:
                        hm0 = V._vit_pose(torch.from_numpy(V.pre_img(crops[0])[0])).numpy()
            kps = np.concatenate(kps, 0).astype(np.float32)
            sdt = O.to_torch_state_dict(sd)
            mine = np.concatenate([O.inference_torch(sdt, shp.depth, shp.num_heads, crops[i]) for i in range(n)], 0)
            print(f'peaked + outliers {variant}/{dataset}: {n} crops x {shp.num_keypoints} joints, confidences {kps[..., 2].min():.3f} .. '
                  f'{kps[..., 2].max():.3f}, oracle-vs-reference keypoints max|d| = {np.abs(mine - kps).max():.3e}')
            np.savez_compressed(os.path.join(HERE, f'peaked_outlier_{variant}_{dataset}.npz'), variant=variant, dataset=dataset, n=n,
                                keypoints=kps, heatmaps0=hm0[:, :16].astype(np.float32))

    # ------------------------------------------------------- full-batch goldens
    # The BASELINE configurations at their own batch sizes, peaked checkpoint, EVERY crop through the reference's own per-crop path
    # (`VitInference._inference_torch`): the device's 256- / 128- / 64- / 512-crop batches are compared joint by joint with these.
    # (minutes of CPU time: ViTPose-H is ~1 s per crop here)
    if want('fullbatch'):
        import time
        from cases import fullbatch_plan, fullbatch_crops
        for variant, dataset, n in fullbatch_plan():
            shp = model_shape(variant, dataset)
            sd = synthetic_state_dict(shp, seed=0, peaked=True)
            V = build_ref(VitInference, ViTPose, dyn_model_import, dataset, variant, sd)
            crops = fullbatch_crops(n)
            t0 = time.time()
            with torch.no_grad():
                kps = np.concatenate([V._inference_torch(crops[i]) for i in range(n)], 0).astype(np.float32)
            print(f'full batch {variant}/{dataset}: {n} crops x {shp.num_keypoints} joints through the reference in {time.time() - t0:.0f} s, '
                  f'confidences {kps[..., 2].min():.3f} .. {kps[..., 2].max():.3f}', flush=True)
            np.savez_compressed(os.path.join(HERE, f'full_{variant}_{dataset}_{n}.npz'), variant=variant, dataset=dataset, n=n, keypoints=kps)

    if want('content'):
        # CONTENT-DEPENDENT checkpoint (round 6; cases.content_state_dict: random backbone + deconvs, ridge-fitted final layer stored as a
        # fixture by fit_content_readout.py): the keypoint LOCATIONS are a function of the crop through all L blocks.  64 crops x every
        # BASELINE model through the reference, crop by crop; the blob centres are stored beside the keypoints so that the tests can
        # show the locations track the content.
        import time
        from cases import content_crops, content_plan, content_state_dict
        for variant, dataset, n in content_plan():
            shp, sd = content_state_dict(variant, dataset)
            V = build_ref(VitInference, ViTPose, dyn_model_import, dataset, variant, sd)
            crops, blobs = content_crops(n)
            t0 = time.time()
            with torch.no_grad():
                kps = np.concatenate([V._inference_torch(crops[i]) for i in range(n)], 0).astype(np.float32)
                # the reference's own heatmaps once more (its pre_img + its model).  A linear read-out of random features gives noisy blobs: 5-10 % of the joints
                # have further pixels -- neighbours of the arg-max or a second maximum 2-3 pixels away -- within 3e-3 of the maximum, i.e. the REFERENCE decides
                # its arg-max by less than its values may differ under `confidences within 1e-3`, and on such a surface the DARK step from the runner-up pixel
                # lands up to 1 px from the step from the winner.  Stored per joint: the four best pixels behind the arg-max with their
                # distance below the maximum, and the keypoint the reference's own decode (post_dark_udp + transform_preds, the calls keypoints_from_heatmaps
                # makes) gives when started from each
                from easy_ViTPose.vit_utils.top_down_eval import post_dark_udp as ref_dark
                from easy_ViTPose.vit_utils.post_processing.post_transforms import transform_preds as ref_tp
                K, NALT = shp.num_keypoints, 4
                alt_yx = np.zeros((n, K, NALT, 2), np.float32)
                alt_margin = np.zeros((n, K, NALT), np.float32)
                cond_px = np.zeros((n, K), np.float32)
                selfcheck = 0.0
                for i in range(n):
                    hm = V._vit_pose(torch.from_numpy(V.pre_img(crops[i])[0])).numpy()
                    for k in range(K):
                        h = hm[0, k]
                        order = np.argsort(-h.ravel(), kind='stable')[:NALT + 1]
                        for a, flat in enumerate(order):          # a = 0: the arg-max itself (self-check against _inference_torch)
                            y0, x0 = divmod(int(flat), h.shape[1])
                            c = ref_dark(np.array([[[x0, y0]]], dtype=np.float32), h[None, None].copy(), kernel=11)
                            xy = ref_tp(c[0], np.array([192 // 2, 256 // 2]), np.array([192, 256]), [48, 64], use_udp=True)[0]
                            if a == 0:
                                selfcheck = max(selfcheck, float(np.abs(xy[::-1] - kps[i, k, :2]).max()))
                            else:
                                alt_yx[i, k, a - 1] = xy[::-1]
                                alt_margin[i, k, a - 1] = h.max() - h[y0, x0]
                    # ... and how far the reference's OWN keypoints move when its heatmaps are perturbed by what `confidences within 1e-3` lets an implementation
                    # differ by (white noise, sigma 3e-4 = the rms error of the fp16 path; 8 seeded draws through the reference's postprocess): where that
                    # exceeds the coordinate tolerance the two tolerances of the north_star contradict each other for ANY implementation that is not
                    # bit-identical (an isolated noise spike as arg-max, a flat top: the DARK step divides by a near-singular Hessian)
                    rng_c = np.random.default_rng(1000 + i)
                    base = V.postprocess(hm.copy(), 192, 256)[0, :, :2]
                    for _ in range(8):
                        pert = hm + rng_c.normal(0.0, 3e-4, size=hm.shape).astype(np.float32)
                        cond_px[i] = np.maximum(cond_px[i], np.abs(V.postprocess(pert, 192, 256)[0, :, :2] - base).max(-1))
                assert selfcheck < 1e-4, f'decode from the arg-max differs from _inference_torch by {selfcheck}'
            # oracle-vs-reference on the first crops (the same check tests/test_oracle_golden.py repeats against the stored outputs)
            from oracle import vitpose_cpu as O
            sdt = O.to_torch_state_dict(sd)
            okp = np.concatenate([O.inference_torch(sdt, shp.depth, shp.num_heads, crops[i]) for i in range(4)])
            K = shp.num_keypoints
            want_yx = np.stack([blobs[:, np.arange(K) % 3, 0], blobs[:, np.arange(K) % 3, 1]], -1)
            loc = np.hypot(kps[..., 0] - want_yx[..., 0], kps[..., 1] - want_yx[..., 1])
            print(f'content {variant}/{dataset}: {n} crops x {K} joints through the reference in {time.time() - t0:.0f} s, confidences '
                  f'{kps[..., 2].min():.3f} .. {kps[..., 2].max():.3f}, distance to the blob centres median {np.median(loc):.2f} px / p90 '
                  f'{np.percentile(loc, 90):.2f} px, oracle-vs-reference max|d| = {np.abs(okp - kps[:4]).max():.3e}; runner-up pixel within 3e-3 of the maximum on '
                  f'{int((alt_margin[..., 0] < 3e-3).sum())} of {alt_margin[..., 0].size} joints (min {alt_margin[..., 0].min():.2e}); the reference '
                  f'moves its own keypoint by > 0.25 px under a 3e-4 heatmap perturbation on {int((cond_px > 0.25).sum())} joints (max {cond_px.max():.2f} px)', flush=True)
            np.savez_compressed(os.path.join(HERE, f'full_content_{variant}_{dataset}.npz'), variant=variant, dataset=dataset, n=n, keypoints=kps,
                                blobs=blobs.astype(np.float32), alt_yx=alt_yx, alt_margin=alt_margin, cond_px=cond_px)


if __name__ == '__main__':
    main()
