#!/usr/bin/env python3
"""Analysis script (not a test): where does the fp16 path's heatmap error come from?

Emulates on the CPU the device pipeline's rounding points (16-bit operands / stored activations, fp32
accumulation and residual stream) one group at a time and reports each group's share of the heatmap error
variance against the fp32 oracle.  Used to decide which stages deserve more precision.

    python tests/precision_budget.py [--variant s] [--crops 4] [--dtype fp16]
    python tests/precision_budget.py --full b/coco            # confidence error per rounding group on a full-batch golden workload (peaked checkpoint)
    python tests/precision_budget.py --content b/coco         # the same on the CONTENT-DEPENDENT checkpoint (round 6): confidence AND coordinate error, encoder vs head
    ... --gelu16                                               # adds the group `hid_pk16`: mlp.fc1's GELU evaluated in PACKED fp16 arithmetic
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easy_vitpose_amd.configs import model_shape
from easy_vitpose_amd.synth import synthetic_crops, synthetic_state_dict
from oracle import vitpose_cpu as O

ap = argparse.ArgumentParser()
ap.add_argument('--variant', default='s')
ap.add_argument('--crops', type=int, default=4)
ap.add_argument('--dtype', default='fp16')
ap.add_argument('--full', default='', help="e.g. b/coco: the CONFIDENCE error (heatmap value at the arg-max) per rounding group on the first --crops crops of "
                "that BASELINE configuration's full-batch golden workload (peaked checkpoint, tests/golden/cases.fullbatch_crops)")
ap.add_argument('--content', default='', help='e.g. b/coco: confidence and coordinate error per rounding group on the content-dependent checkpoint (cases.content_state_dict)')
ap.add_argument('--gelu16', action='store_true', help='add the group hid_pk16: the GELU polynomial, exponent and final multiply-add in fp16 arithmetic (what v_pk_fma_f16 would compute)')
args = ap.parse_args()
DT = torch.float16 if args.dtype == 'fp16' else torch.bfloat16

ENCODER = ['patch_in', 'patch_w', 'ln_out', 'qkv_w', 'qkv_out', 'attn_p', 'attn_out', 'proj_w', 'fc1_w', 'hid', 'fc2_w']
HEAD = ['lastnorm_out', 'd1_w', 'd1_out', 'd2_w', 'd2_out', 'final_w']
GROUPS = ['patch_in', 'patch_w', 'ln_out', 'qkv_w', 'qkv_out', 'attn_p', 'attn_out', 'proj_w', 'fc1_w', 'hid', 'fc2_w',
          'lastnorm_out', 'd1_w', 'd1_out', 'd2_w', 'd2_out', 'final_w']


def gelu_pk16(x):
    """common.h::gelu_core with every operation after the fp32 LayerNorm fold in fp16: x rounded to fp16, the degree-4 polynomial, the exponent argument and the
    final multiply-add as fp16 fused multiply-adds (product exact in fp32, one rounding to fp16), exp2 in fp16 (v_exp_f16: result rounded to fp16)."""
    h16 = lambda t: t.to(torch.float16).float()
    fma = lambda a, b, c: h16(a * b + c)
    xh = h16(x)
    a = xh.abs()
    q = fma(a, h16(torch.tensor(5.204574411e-04)), h16(torch.tensor(-7.397505390e-03)))
    q = fma(q, a, h16(torch.tensor(5.256122897e-02)))
    q = fma(q, a, h16(torch.tensor(4.592546873e-01)))
    q = fma(q, a, h16(torch.tensor(1.151091354e+00)))
    e = h16(torch.exp2(fma(-q, a, torch.tensor(-1.0))))
    return fma(-a, e, torch.clamp(xh, min=0.0))


def fwd(sd, x, depth, heads, on):
    def r(t, g):
        return t.to(DT).float() if g in on else t
    w = sd['backbone.patch_embed.proj.weight']
    x = F.conv2d(r(x, 'patch_in'), r(w, 'patch_w'), sd['backbone.patch_embed.proj.bias'], stride=16, padding=2)
    B, D, Hp, Wp = x.shape
    x = x.view(B, D, Hp * Wp).transpose(1, 2)
    pos = sd['backbone.pos_embed']
    x = x + pos[:, 1:] + pos[:, :1]
    hd = D // heads
    for i in range(depth):
        p = f'backbone.blocks.{i}.'
        y = r(F.layer_norm(x, (D,), sd[p + 'norm1.weight'], sd[p + 'norm1.bias'], eps=1e-6), 'ln_out')
        qkv = r(F.linear(y, r(sd[p + 'attn.qkv.weight'], 'qkv_w'), sd[p + 'attn.qkv.bias']), 'qkv_out')
        qkv = qkv.reshape(B, Hp * Wp, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        s = (q @ k.transpose(-2, -1)) * hd ** -0.5
        e = torch.exp(s - s.max(-1, keepdim=True).values)
        y = (r(e, 'attn_p') @ v) / e.sum(-1, keepdim=True)          # device: P un-normalised in 16 bit, 1/l in fp32
        y = r(y.transpose(1, 2).reshape(B, Hp * Wp, D), 'attn_out')
        x = x + F.linear(y, r(sd[p + 'attn.proj.weight'], 'proj_w'), sd[p + 'attn.proj.bias'])
        y = r(F.layer_norm(x, (D,), sd[p + 'norm2.weight'], sd[p + 'norm2.bias'], eps=1e-6), 'ln_out')
        y = F.linear(y, r(sd[p + 'mlp.fc1.weight'], 'fc1_w'), sd[p + 'mlp.fc1.bias'])
        y = gelu_pk16(y) if 'hid_pk16' in on else r(F.gelu(y), 'hid')
        x = x + F.linear(y, r(sd[p + 'mlp.fc2.weight'], 'fc2_w'), sd[p + 'mlp.fc2.bias'])
    x = r(F.layer_norm(x, (D,), sd['backbone.last_norm.weight'], sd['backbone.last_norm.bias'], eps=1e-6), 'lastnorm_out')
    x = x.permute(0, 2, 1).reshape(B, D, 16, 12)
    h = 'keypoint_head.deconv_layers.'
    for idx, tag in ((0, 'd1'), (3, 'd2')):
        # the device folds BatchNorm (eval) into the deconv weights before rounding them
        sc = sd[f'{h}{idx + 1}.weight'] / torch.sqrt(sd[f'{h}{idx + 1}.running_var'] + 1e-5)
        wf = r(sd[f'{h}{idx}.weight'] * sc.view(1, -1, 1, 1), tag + '_w')
        bf = sd[f'{h}{idx + 1}.bias'] - sd[f'{h}{idx + 1}.running_mean'] * sc
        x = r(F.relu(F.conv_transpose2d(x, wf, bf, stride=2, padding=1)), tag + '_out')
    return F.conv2d(x, r(sd['keypoint_head.final_layer.weight'], 'final_w'), sd['keypoint_head.final_layer.bias'])


if args.full:
    # which rounding point owns the 9.6e-4 confidence error of the full-batch goldens?  Confidence = the heatmap's maximum per joint
    # (top_down_eval.py:82-114), so its error is the heatmap error AT the peak: evaluated there, per joint, one rounding group at a time.
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    from cases import fullbatch_crops, fullbatch_plan
    variant, dataset = args.full.split('/')
    n_full = {(v, d): n for v, d, n in fullbatch_plan()}[(variant, dataset)]
    with torch.no_grad():
        shp = model_shape(variant, dataset)
        sd = O.to_torch_state_dict(synthetic_state_dict(shp, 0, peaked=True))
        crops = fullbatch_crops(n_full)[:args.crops]
        x = torch.from_numpy(np.concatenate([O.pre_img(c)[0] for c in crops]))

        def conf(on):
            hm = torch.cat([fwd(sd, x[i:i + 4], shp.depth, shp.num_heads, on) for i in range(0, len(x), 4)])
            return hm.flatten(2).max(-1).values.numpy().astype(np.float64)
        ref = conf(set())
        full = conf(set(GROUPS)) - ref
        print(f'{args.full}: {full.size} joints of the first {len(x)} crops of the {n_full}-crop golden workload, peaked checkpoint, {args.dtype}: confidences '
              f'{ref.min():.3f} .. {ref.max():.3f}; all roundings on: confidence error rms {np.sqrt((full ** 2).mean()):.3e} max {np.abs(full).max():.3e}')
        rows = []
        for g in GROUPS:
            e = conf({g}) - ref
            rows.append((float((e ** 2).mean()), float(np.abs(e).max()), g))
        ssum = sum(v for v, _, _ in rows)
        for v, mx, g in sorted(rows, reverse=True):
            print(f'  {g:14s} rms {v ** 0.5:.3e}  max {mx:.3e}  {100 * v / ssum:5.1f} % of the summed variance')
        print(f'  sum of single-group variances / all-on variance = {ssum / float((full ** 2).mean()):.2f}')
        wsum = sum(v for v, _, g in rows if g.endswith('_w'))
        print(f'  weight roundings {100 * wsum / ssum:.1f} %, activation roundings {100 * (1 - wsum / ssum):.1f} %; head (lastnorm_out .. final_w) '
              f'{100 * sum(v for v, _, g in rows if g in ("lastnorm_out", "d1_w", "d1_out", "d2_w", "d2_out", "final_w")) / ssum:.1f} %')
    sys.exit(0)

if args.content:
    # on the content-dependent checkpoint the keypoint LOCATIONS depend on the crop through all L blocks -- which share of the confidence
    # and of the COORDINATE error is the encoder's?  Per rounding group: decode (arg-max + DARK/UDP, oracle) of the emulated heatmaps against the fp32 ones.
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    from cases import content_crops, content_state_dict
    variant, dataset = args.content.split('/')
    with torch.no_grad():
        shp, sd_np = content_state_dict(variant, dataset)
        sd = O.to_torch_state_dict(sd_np)
        crops, _ = content_crops(64)
        crops = crops[:args.crops]
        x = torch.from_numpy(np.concatenate([O.pre_img(c)[0] for c in crops]))

        def kp(on):
            hm = torch.cat([fwd(sd, x[i:i + 4], shp.depth, shp.num_heads, on) for i in range(0, len(x), 4)]).numpy()
            return O.decode_per_crop(hm).astype(np.float64)
        ref = kp(set())
        groups = GROUPS + (['hid_pk16'] if args.gelu16 else [])
        allon = kp(set(GROUPS)) - ref
        print(f'{args.content}: {ref[..., 2].size} joints of the first {len(x)} crops of the content-dependent golden workload, {args.dtype}: confidences {ref[..., 2].min():.3f} .. '
              f'{ref[..., 2].max():.3f}; all roundings on: confidence error rms {np.sqrt((allon[..., 2] ** 2).mean()):.3e} max {np.abs(allon[..., 2]).max():.3e}, coordinate error '
              f'rms {np.sqrt((allon[..., :2] ** 2).mean()):.4f} px max {np.abs(allon[..., :2]).max():.4f} px')
        rows = []
        for g in groups:
            e = kp({g}) - ref
            rows.append((float((e[..., 2] ** 2).mean()), float(np.abs(e[..., 2]).max()), float((e[..., :2] ** 2).mean()), float(np.abs(e[..., :2]).max()), g))
        csum = sum(r[0] for r in rows if r[4] in GROUPS)
        xsum = sum(r[2] for r in rows if r[4] in GROUPS)
        for cv, cm, xv, xm, g in sorted(rows, reverse=True):
            print(f'  {g:14s} confidence rms {cv ** 0.5:.3e} max {cm:.3e} ({100 * cv / csum:5.1f} %)   coordinates rms {xv ** 0.5:.4f} px max {xm:.4f} px ({100 * xv / xsum:5.1f} %)')
        enc_c = sum(r[0] for r in rows if r[4] in ENCODER) / csum
        enc_x = sum(r[2] for r in rows if r[4] in ENCODER) / xsum
        print(f'  ENCODER (patch embed .. mlp.fc2) share of the summed variance: confidence {100 * enc_c:.1f} %, coordinates {100 * enc_x:.1f} %; head {100 * (1 - enc_c):.1f} % / {100 * (1 - enc_x):.1f} %')
    sys.exit(0)

with torch.no_grad():
    shp = model_shape(args.variant, 'coco')
    sd = O.to_torch_state_dict(synthetic_state_dict(shp, 0))
    crops = synthetic_crops(args.crops, 8, 'noise')
    x = torch.from_numpy(np.concatenate([O.pre_img(c)[0] for c in crops]))
    ref = fwd(sd, x, shp.depth, shp.num_heads, set())
    full = fwd(sd, x, shp.depth, shp.num_heads, set(GROUPS)) - ref
    tot = float((full ** 2).mean())
    print(f'variant {args.variant} {args.dtype}: heatmap std {float(ref.std()):.3f}; all roundings: rms {tot ** 0.5:.3e} max {float(full.abs().max()):.3e}')
    rows = []
    for g in GROUPS + (['hid_pk16'] if args.gelu16 else []):
        e = fwd(sd, x, shp.depth, shp.num_heads, {g}) - ref
        rows.append((float((e ** 2).mean()), g))
    s = sum(v for v, _ in rows)
    for v, g in sorted(rows, reverse=True):
        print(f'  {g:14s} rms {v ** 0.5:.3e}  {100 * v / s:5.1f} % of the summed variance')
    print(f'  sum of single-group variances / all-on variance = {s / tot:.2f}')
