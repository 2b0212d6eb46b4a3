#!/usr/bin/env python3
"""Analysis script (not a test): BASELINE config 5 -- ViTPose-B / AP-10K with fp8 (OCP e4m3) encoder GEMM operands.

Emulates on the CPU, bit-exactly in the operand rounding (torch.float8_e4m3fn, round to nearest even, saturating), what an
`mfma_scale_f32_16x16x128_f8f6f4` encoder would compute, on top of the fp16 device pipeline's other rounding points:

  W8      qkv / proj / fc1 / fc2 weights in e4m3 with one fp32 scale per OUTPUT CHANNEL (max |w_row| / 448), activations fp16
          (no fp8 MFMA exists for fp16 x fp8 operands: this variant only saves weight bytes -- irrelevant at batch 512)
  W8A8    the same + the A operand of those four GEMMs in e4m3 with one fp32 scale per TOKEN ROW (dynamic, max |row| / 448):
          the only variant that runs on the 5 PFLOP/s fp8 pipe
  W8A8-t  W8A8 with one static per-TENSOR activation scale (what a producer epilogue can apply without a row reduction)

against the fp32 oracle, for the random checkpoint (noise-like maps, std 0.3) and the peaked one (one blob per joint), and
prints heatmap rms / max error, confidence error and -- peaked maps -- coordinate error over all joints.

    python tests/fp8_budget.py [--crops 4] > profiles/fp8_study_r2.txt

The operand quantisation used here (`q8_rows`) is confirmed on the hardware by tests/test_gpu_ops.py::test_fp8_probe_confirms_the_emulation
(device codes == torch.float8_e4m3fn codes bit for bit; v_mfma_f32_16x16x128_f8f6f4 product == the emulation's product).
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

F8 = torch.float8_e4m3fn
F8MAX = 448.0


def r16(t):
    return t.to(torch.float16).float()


def q8_rows(t):
    """e4m3 with one fp32 scale per row of the last-but-one dimension (rows = output channels of W / tokens of A)."""
    s = t.abs().amax(-1, keepdim=True).clamp_min(1e-12) / F8MAX
    return (t / s).to(F8).float() * s


def q8_tensor(t, s):
    return (t / s).clamp(-F8MAX, F8MAX).to(F8).float() * s


def q8_mx(t):
    """MXFP8 as the fp8 mode's kernels write it (csrc/mx8.h): e4m3 codes + one power-of-two (E8M0) scale per 32 consecutive k, the block's amax scaled into [128, 256)."""
    shp = t.shape
    b = t.reshape(*shp[:-1], shp[-1] // 32, 32)
    amax = b.abs().amax(-1, keepdim=True).clamp_min(2.0 ** -100)
    s = torch.exp2(torch.floor(torch.log2(amax)) - 7.0)          # amax / s in [128, 256)
    return ((b / s).to(F8).float() * s).reshape(shp)


# GEMM family of each A-operand key / weight: the round-5 table "which single family on MXFP8 keeps the confidences" 
FAMILY_OF = {'ln1': 'qkv', 'attn': 'proj', 'ln2': 'fc1', 'hid': 'fc2'}


def fwd(sd, x, depth, heads, mode, act_scales=None, record=None, families=None):
    """mode: 'fp32', 'fp16' (device pipeline), 'w8', 'w8a8', 'w8a8t', 'mx' (the shipped fp8 mode's formats); families: the GEMM families that run on
    fp8 operands (default all four), the others keep fp16"""
    lo = mode != 'fp32'
    r = (lambda t: r16(t)) if lo else (lambda t: t)
    fams = set(FAMILY_OF.values()) if families is None else set(families)
    def wq(w, fam):      # encoder GEMM weight
        return q8_rows(w) if (mode in ('w8', 'w8a8', 'w8a8t', 'mx') and fam in fams) else r(w)
    def aq(a, key):  # encoder GEMM A operand
        if record is not None:
            record[key] = max(record.get(key, 0.0), float(a.abs().max()))
        if FAMILY_OF[key] not in fams:
            return r(a)
        if mode == 'w8a8':
            return q8_rows(a)
        if mode == 'w8a8t':
            return q8_tensor(a, act_scales[key] / F8MAX)
        if mode == 'mx':
            return q8_mx(a)
        return r(a)
    w = sd['backbone.patch_embed.proj.weight']
    x = F.conv2d(r(x), r(w), sd['backbone.patch_embed.proj.bias'], stride=16, padding=2)
    B, D, Hp, Wp = x.shape
    x = x.view(B, D, Hp * Wp).transpose(1, 2)
    pos = sd['backbone.pos_embed']
    x = x + pos[:, 1:] + pos[:, :1]
    hd = D // heads
    for i in range(depth):
        p = f'backbone.blocks.{i}.'
        y = aq(F.layer_norm(x, (D,), sd[p + 'norm1.weight'], sd[p + 'norm1.bias'], eps=1e-6), 'ln1')
        qkv = r(F.linear(y, wq(sd[p + 'attn.qkv.weight'], 'qkv'), sd[p + 'attn.qkv.bias']))
        qkv = qkv.reshape(B, Hp * Wp, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        s = (q @ k.transpose(-2, -1)) * hd ** -0.5
        e = torch.exp(s - s.max(-1, keepdim=True).values)
        y = (r(e) @ v) / e.sum(-1, keepdim=True)
        y = aq(y.transpose(1, 2).reshape(B, Hp * Wp, D), 'attn')
        x = x + F.linear(y, wq(sd[p + 'attn.proj.weight'], 'proj'), sd[p + 'attn.proj.bias'])
        y = aq(F.layer_norm(x, (D,), sd[p + 'norm2.weight'], sd[p + 'norm2.bias'], eps=1e-6), 'ln2')
        y = aq(F.gelu(F.linear(y, wq(sd[p + 'mlp.fc1.weight'], 'fc1'), sd[p + 'mlp.fc1.bias'])), 'hid')
        x = x + F.linear(y, wq(sd[p + 'mlp.fc2.weight'], 'fc2'), sd[p + 'mlp.fc2.bias'])
    x = r(F.layer_norm(x, (D,), sd['backbone.last_norm.weight'], sd['backbone.last_norm.bias'], eps=1e-6))
    x = x.permute(0, 2, 1).reshape(B, D, 16, 12)
    h = 'keypoint_head.deconv_layers.'
    for idx in (0, 3):
        sc = sd[f'{h}{idx + 1}.weight'] / torch.sqrt(sd[f'{h}{idx + 1}.running_var'] + 1e-5)
        wf = r(sd[f'{h}{idx}.weight'] * sc.view(1, -1, 1, 1))
        bf = sd[f'{h}{idx + 1}.bias'] - sd[f'{h}{idx + 1}.running_mean'] * sc
        x = r(F.relu(F.conv_transpose2d(x, wf, bf, stride=2, padding=1)))
    return F.conv2d(x, sd['keypoint_head.final_layer.weight'], sd['keypoint_head.final_layer.bias'])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--crops', type=int, default=4)
    ap.add_argument('--variant', default='b')
    ap.add_argument('--dataset', default='ap10k')
    ap.add_argument('--families', action='store_true', help='MXFP8 on ONE GEMM family at a time, on the first --crops crops of the configuration\'s full-batch golden '
                    'workload (peaked checkpoint): which family alone keeps the confidences within 1e-3?')
    args = ap.parse_args()
    torch.set_num_threads(16)
    if args.families:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
        from cases import fullbatch_crops, fullbatch_plan
        n_full = {(v, d): n for v, d, n in fullbatch_plan()}[(args.variant, args.dataset)]
        with torch.no_grad():
            shp = model_shape(args.variant, args.dataset)
            sd = O.to_torch_state_dict(synthetic_state_dict(shp, 0, peaked=True))
            crops = fullbatch_crops(n_full)[:args.crops]
            x = torch.from_numpy(np.concatenate([O.pre_img(c)[0] for c in crops]))
            def run(mode, fams=None):
                return np.concatenate([fwd(sd, x[i:i + 4], shp.depth, shp.num_heads, mode, families=fams).numpy() for i in range(0, len(x), 4)])
            ref = run('fp32')
            ref_kp = O.decode_per_crop(ref)
            print(f'# ViTPose-{args.variant.upper()} / {args.dataset}: first {len(x)} crops of the {n_full}-crop golden workload, peaked checkpoint, {ref_kp[..., 2].size} joints; errors against the fp32 oracle')
            print(f'{"operands":34s} {"conf max":>10s} {"conf rms":>10s} {"joints > 1e-3":>14s} {"coord max px":>13s}')
            rows = [('fp16 everywhere (shipped default)', 'fp16', None)] + [(f'MXFP8 {f} only', 'mx', [f]) for f in ('qkv', 'proj', 'fc1', 'fc2')] + \
                   [('MXFP8 fc1 + fc2', 'mx', ['fc1', 'fc2']), ('MXFP8 qkv + proj', 'mx', ['qkv', 'proj']), ('MXFP8 all four (the fp8 mode)', 'mx', None),
                    ('e4m3 WEIGHTS only, all four', 'w8', None)]
            for name, mode, fams in rows:
                kp = O.decode_per_crop(run(mode, fams))
                dc = np.abs(kp[..., 2] - ref_kp[..., 2])
                dp = np.abs(kp[..., :2] - ref_kp[..., :2]).max(-1)
                print(f'{name:34s} {dc.max():10.3e} {np.sqrt((dc ** 2).mean()):10.3e} {int((dc > 1e-3).sum()):8d} / {dc.size:<4d} {dp.max():13.3f}', flush=True)
        return
    with torch.no_grad():
        shp = model_shape(args.variant, args.dataset)
        crops = np.concatenate([synthetic_crops(args.crops // 2, 21, 'blobs'), synthetic_crops(args.crops - args.crops // 2, 22, 'noise')])
        x = torch.from_numpy(np.concatenate([O.pre_img(c)[0] for c in crops]))
        print(f'# ViTPose-{args.variant.upper()} / {args.dataset} (K = {shp.num_keypoints}), {args.crops} crops; errors against the fp32 oracle; tolerance: 1e-3 confidence, 0.5 px')
        for peaked in (False, True):
            sd = O.to_torch_state_dict(synthetic_state_dict(shp, 0, peaked=peaked))
            ref = fwd(sd, x, shp.depth, shp.num_heads, 'fp32').numpy()
            ref_kp = O.decode_per_crop(ref)
            rec = {}
            fwd(sd, x, shp.depth, shp.num_heads, 'fp16', record=rec)
            print(f'\n== {"peaked" if peaked else "random"} checkpoint: heatmap std {ref.std():.3f}, confidences {ref_kp[..., 2].min():.2f} .. {ref_kp[..., 2].max():.2f}; '
                  f'max |A operand|: ' + ', '.join(f'{k} {v:.1f}' for k, v in rec.items()))
            print(f'{"mode":8s} {"heatmap rms":>12s} {"heatmap max":>12s} {"conf max":>10s} {"conf rms":>10s} {"coord max px":>13s} {"joints > 1e-3":>14s} {"joints > 0.5px":>15s}')
            for mode in ('fp16', 'w8', 'w8a8', 'w8a8t'):
                hm = fwd(sd, x, shp.depth, shp.num_heads, mode, act_scales=rec).numpy()
                kp = O.decode_per_crop(hm)
                e = hm - ref
                dc = np.abs(kp[..., 2] - ref_kp[..., 2])
                dp = np.abs(kp[..., :2] - ref_kp[..., :2]).max(-1)
                coord = f'{dp.max():13.3f}' if peaked else f'{"(noise maps)":>13s}'
                print(f'{mode:8s} {np.sqrt((e ** 2).mean()):12.3e} {np.abs(e).max():12.3e} {dc.max():10.3e} {np.sqrt((dc ** 2).mean()):10.3e} {coord} '
                      f'{int((dc > 1e-3).sum()):8d} / {dc.size:<4d} {(int((dp > 0.5).sum()) if peaked else 0):9d} / {dp.size:<4d}')


if __name__ == '__main__':
    main()
