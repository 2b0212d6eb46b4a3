"""CPU: host-side logic and the C-ABI surface (no compute calls without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from easy_vitpose_amd import _capi as capi
from easy_vitpose_amd import configs
from easy_vitpose_amd.cropprep import crop_params, prepare_crops_host, resize_linear_u8
from easy_vitpose_amd.inference import pad_image
from easy_vitpose_amd.parallel import shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shape_table_matches_reference_configs():
    # ViTPose_common.py:65-195 and the per-dataset out_channels
    assert configs.VARIANTS == {'s': (384, 12, 12), 'b': (768, 12, 12), 'l': (1024, 24, 16), 'h': (1280, 32, 16)}
    assert configs.DATASET_KEYPOINTS['coco'] == 17 and configs.DATASET_KEYPOINTS['coco_25'] == 25
    assert configs.DATASET_KEYPOINTS['wholebody'] == 133 and configs.DATASET_KEYPOINTS['ap10k'] == 17
    s = configs.model_shape('h', 'wholebody')
    assert (s.head_dim, s.mlp_dim, s.num_keypoints) == (80, 5120, 133)
    # algorithmic GFLOP per person, SURVEY.md 8(d) / BASELINE.md
    for (v, d), gf in {('s', 'coco'): 11.188, ('b', 'coco'): 37.046, ('l', 'coco_25'): 123.151, ('h', 'wholebody'): 251.842}.items():
        assert abs(configs.model_shape(v, d).gflop_per_person() - gf) < 2e-3
    with pytest.raises(AssertionError):
        configs.model_shape('x', 'coco')
    with pytest.raises(AssertionError):
        configs.model_shape('b', 'imagenet')


def test_infer_dataset_by_path_like_reference():
    assert configs.infer_dataset_by_path('/a/b/vitpose-b-coco_25.pth') == 'coco_25'
    assert configs.infer_dataset_by_path('vitpose-h-wholebody.pth') == 'wholebody'
    with pytest.raises(ValueError):
        configs.infer_dataset_by_path('model.pth')


def test_pad_image_contract():
    img = np.arange(10 * 30 * 3, dtype=np.uint8).reshape(10, 30, 3)      # wide -> pad rows
    out, (left, top) = pad_image(img, 3 / 4)
    assert out.shape == (40, 30, 3) and (left, top) == (0, 15)
    assert np.array_equal(out[15:25], img) and out[:15].sum() == 0 and out[25:].sum() == 0
    img = np.ones((40, 10, 3), np.uint8)                                 # tall -> pad columns
    out, (left, top) = pad_image(img, 3 / 4)
    assert out.shape == (40, 30, 3) and (left, top) == (10, 0) and out[:, 10:20].min() == 1
    img = np.ones((256, 192, 3), np.uint8)
    out, pads = pad_image(img, 3 / 4)
    assert out.shape == (256, 192, 3) and pads == (0, 0)


def test_resize_restates_opencv_fixed_point_bilinear():
    img = np.random.default_rng(0).integers(0, 256, (256, 192, 3), dtype=np.uint8)
    assert resize_linear_u8(img, (192, 256)) is img
    # exactly 2x: OpenCV switches INTER_LINEAR to the 2x2 box average, (sum + 2) >> 2
    big = np.random.default_rng(1).integers(0, 256, (512, 384, 3), dtype=np.uint8)
    ref = ((big.astype(np.int64).reshape(256, 2, 192, 2, 3).sum(axis=(1, 3)) + 2) >> 2).astype(np.uint8)
    assert np.array_equal(resize_linear_u8(big, (192, 256)), ref)
    assert (resize_linear_u8(np.full((100, 75, 3), 77, np.uint8), (192, 256)) == 77).all()
    # generic scale: within one grey level of exact (float64) half-pixel-centre bilinear interpolation
    src = np.random.default_rng(2).integers(0, 256, (300, 225, 3), dtype=np.uint8)
    ys = (np.arange(256) + 0.5) * (300 / 256) - 0.5; xs = (np.arange(192) + 0.5) * (225 / 192) - 0.5
    y0 = np.clip(np.floor(ys).astype(int), 0, 299); x0 = np.clip(np.floor(xs).astype(int), 0, 224)
    y1 = np.clip(y0 + 1, 0, 299); x1 = np.clip(x0 + 1, 0, 224)
    fy = np.clip(ys - np.floor(ys), 0, 1)[:, None, None] * (ys >= 0)[:, None, None]
    fx = np.clip(xs - np.floor(xs), 0, 1)[None, :, None] * (xs >= 0)[None, :, None]
    f = src.astype(np.float64)
    exact = (f[y0][:, x0] * (1 - fx) + f[y0][:, x1] * fx) * (1 - fy) + (f[y1][:, x0] * (1 - fx) + f[y1][:, x1] * fx) * fy
    assert np.abs(resize_linear_u8(src, (192, 256)).astype(np.float64) - exact).max() <= 1.0


def test_crop_params_follow_reference_geometry():
    frame = np.random.default_rng(3).integers(0, 256, (480, 640, 3), dtype=np.uint8)
    boxes = np.array([[60, 110, 232, 346, 0.9], [0, 0, 50, 300, 0.5], [600, 400, 640, 480, 0.7]])
    p = crop_params(boxes, frame.shape[:2])
    assert p[0].tolist() == [50, 100, 192, 256, 0, 0, 192, 256]          # +10 px, already 3:4
    assert p[1].tolist() == [0, 0, 60, 310, 86, 0, 232, 310]            # clipped at 0, tall -> padded left/right
    assert p[2].tolist() == [590, 390, 50, 90, 8, 0, 67, 90]            # clipped at the frame border
    crops = prepare_crops_host(frame, p)
    assert crops.shape == (3, 256, 192, 3)
    assert np.array_equal(crops[0], frame[100:356, 50:242])             # identity resize of the exact box
    # same as the reference's two steps: pad_image then resize
    for i, (x0, y0, cw, ch, left, top, pw, ph) in enumerate(p):
        padded, (l, t) = pad_image(frame[y0:y0 + ch, x0:x0 + cw], 3 / 4)
        assert (l, t) == (left, top) and padded.shape[:2] == (ph, pw)
        assert np.array_equal(resize_linear_u8(padded, (192, 256)), crops[i])


def test_pad_image_matches_reference_golden():
    """Outputs of the reference's own pad_image (vit_utils/inference.py:41-70) on seeded crops, made by
    tests/golden/make_golden.py: padded shape, pads, pixel sum, and the source kept intact inside the padding."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    from cases import pad_shapes
    rows = np.load(os.path.join(ROOT, 'tests', 'golden', 'pad_image.npz'))['rows']
    assert len(rows) == len(pad_shapes())
    for i, ((h, w), row) in enumerate(zip(pad_shapes(), rows)):
        img = np.random.default_rng(100 + i).integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        out, (left, top) = pad_image(img, 3 / 4)
        assert [h, w, out.shape[0], out.shape[1], left, top, int(out.astype(np.int64).sum())] == row[:7].tolist()
        assert int((out[top:top + h, left:left + w] != img).sum()) == row[7] == 0


def test_crop_geometry_matches_reference_frame_golden():
    """The reference's VitInference.inference box loop (inference.py:252-262) run by make_golden.py on the frame of
    tests/golden/cases.frame_case: threshold 0.35, round, +10 px, clip -> the padded boxes it saved in _tracker_res."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    from cases import frame_case
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'frame_inference.npz'))
    frame, boxes = frame_case()
    det = boxes[boxes[:, 4] > 0.35]
    p = crop_params(det[:, :4].round().astype(int), frame.shape[:2], 10)
    mine = np.stack([p[:, 0], p[:, 1], p[:, 0] + p[:, 2], p[:, 1] + p[:, 3]], 1)
    assert np.array_equal(mine, g['padded_boxes'])
    assert g['scores'].tolist() == det[:, 4].tolist() and g['ids'].tolist() == [0, 1, 2]


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 8, 64, 65, 257):
        for w in (1, 2, 4, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            per = -(-n // w) if n else 0
            assert all(hi - lo <= per for lo, hi in spans)


def test_library_loads_and_exports_every_header_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'vitpose_hip.h')).read()
    declared = sorted(set(re.findall(r'VP_API\s+[\w\s\*]+?\b(vp_\w+)\s*\(', hdr)))
    assert len(declared) >= 15
    lib = capi.load_library()
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in the header but not exported'
    assert sorted(capi.SYMBOLS) == declared, 'ctypes binding list and header disagree'
    assert lib.vp_abi_version() == 4
    # struct layout the header promises
    assert C.sizeof(capi.vp_config) == 28 and C.sizeof(capi.vp_tensor_desc) == 24
    assert C.sizeof(capi.vp_profile) == capi.VP_PROF_COUNT * 8 * 4


def test_tools_library_exports_the_measurement_entry_points():
    """include/vitpose_hip_tools.h = the -DVP_TOOLS build of the same sources (tools/ only): it exports everything the product
    header declares plus the timeline taps and the fused-kernel timing loop; the PRODUCT library does not carry those."""
    from easy_vitpose_amd.build import build_library, TOOLS_LIB
    hdr = open(os.path.join(ROOT, 'include', 'vitpose_hip_tools.h')).read()
    tools_only = sorted(set(re.findall(r'VP_API\s+[\w\s\*]+?\b(vp_\w+)\s*\(', hdr)))
    # round 6: the timing taps (gemm_bench / bench2 / compare, peak) left the product library too -- it exports parity taps only
    assert tools_only == ['vp_dbg_gemm8_timeline', 'vp_dbg_gemm_bench', 'vp_dbg_gemm_bench2', 'vp_dbg_gemm_compare', 'vp_dbg_gemm_timeline',
                          'vp_dbg_hwid_probe', 'vp_dbg_peak', 'vp_dbg_qkvattn_bench']
    tl = C.CDLL(build_library(tools=True))
    assert os.path.samefile(TOOLS_LIB, tl._name)
    for name in tools_only + list(capi.SYMBOLS):
        assert hasattr(tl, name), name
    prod = capi.load_library()
    for name in tools_only:
        assert not hasattr(prod, name), f'{name} must not be in the product library'


def test_split_k_rule_over_every_batch_size():
    """Round 6: the split-K rule of the residual GEMMs (tile_rules.hip pick_splitk through the host-only tap vp_dbg_splitk_pick) walked over every batch size of
    every model.  Invariants: attn.proj is never split; S in {1, 4}; K divides into S ranges of whole 128-k double blocks; the partial products run on a tile the
    product library instantiates for EPI_PARTIAL; S x tiles fills at most two rounds of workgroups; and the operating points measured in situ
    (profiles/small_batch_r6.txt) keep their choices -- 1-2 crops split, 4+ crops not, ViTPose-H's 7-8 crops on 128 x 128 tiles."""
    lib = capi.load_library()
    var = C.c_int32()
    tile = {31: (32, 64), 12: (64, 64), 1: (128, 128), 11: (192, 128), 15: (128, 64), 20: (192, 128), 30: (64, 64)}
    for D in (384, 768, 1024, 1280):
        for n in range(1, 401):
            M = 192 * n
            assert lib.vp_dbg_splitk_pick(M, D, D, C.byref(var)) == 1, 'attn.proj is never split'
            S = lib.vp_dbg_splitk_pick(M, D, 4 * D, C.byref(var))
            assert S in (1, 4), (D, n, S)
            if S > 1:
                assert (4 * D) % (128 * S) == 0 and var.value in tile and n <= 8, (D, n, S, var.value)
                bm, bn = tile[var.value]
                assert S * -(-M // bm) * -(-D // bn) <= 1024, (D, n)
    pick = lambda D, n: (lib.vp_dbg_splitk_pick(192 * n, D, 4 * D, C.byref(var)), var.value)
    assert pick(768, 1) == (4, 31) and pick(1024, 1) == (4, 31) and pick(1280, 1) == (4, 12) and pick(384, 1) == (4, 31)
    assert pick(768, 2) == (4, 12) and pick(1024, 2) == (4, 12) and pick(1280, 2) == (4, 12) and pick(384, 2)[0] == 1
    for D in (384, 768, 1024):
        for n in (3, 4, 6, 8, 12, 16, 64, 256):
            assert pick(D, n)[0] == 1, (D, n)
    assert pick(1280, 8) == (4, 1) and pick(1280, 7) == (4, 1) and pick(1280, 6)[0] == 1 and pick(1280, 4)[0] == 1 and pick(1280, 12)[0] == 1


def test_gemm2_tile_selection_over_every_batch_size():
    """The rule that picks the 2-phase kernel's tile configuration (tile_rules.hip pick_gemm2_tile, through the host-only tap vp_dbg_gemm2_pick) walked over every
    batch size 1..400 of every model and every GEMM of the path.  Invariants: only configurations the PRODUCT library instantiates; the deep rings (4-stage 64 x 64,
    3-stage 128 x 64, the two-k-blocks-per-barrier configurations) only where ALL their tiles are resident at once (one round) -- except the round-2 long-K case --;
    two k-blocks per barrier only for an even number of k-blocks; the default 192 x 128 tile from 384 tiles on -- and below wherever the 128 x 128 tiles would
    overflow the 512 resident slots --, except the one-round 256 x 256 tile for a wide GEMM with more than 512 such tiles; the 96 x 64 / 128 x 64 / one-round
    192 x 128 ladder of the residual GEMMs beyond 512 tiles of 64 x 64 (round 6); and the operating points measured in rounds 5-6
    (profiles/small_batch_r5.txt, profiles/small_batch_r6.txt) keep their kernels."""
    lib = capi.load_library()
    gm = C.c_int32()
    # Cfg id -> (BM, BN, resident workgroups on 256 CUs, two k-blocks per barrier)
    cfgs = {8: (192, 128, 512, False), 11: (192, 128, 512, False), 1: (128, 128, 512, False), 9: (64, 64, 1280, False), 12: (64, 64, 512, False),
            15: (128, 64, 512, False), 30: (64, 64, 256, True), 31: (32, 64, 512, True), 20: (192, 128, 256, False), 3: (256, 256, 256, False), 41: (96, 64, 512, False)}

    def pick(epi, M, N, K):
        v = lib.vp_dbg_gemm2_pick(epi, M, N, K, C.byref(gm))
        return v, gm.value

    for D in (384, 768, 1024, 1280):
        for n in range(1, 401):
            M = 192 * n
            gemms = [(0, M, 3 * D, D, 1), (1, M, 4 * D, D, 1), (6, M, D, D, 1), (6, M, D, 4 * D, 1), (7, M, D, 768, 1),
                     (4, M, 256, 4 * D, 4), (4, 4 * M, 256, 1024, 4), (5, 16 * M, 64, 256, 1)]
            for epi, m, N, K, par in gemms:
                v, g = pick(epi, m, N, K)
                assert v in cfgs, (D, n, epi, v)
                bm, bn, slots, two = cfgs[v]
                tiles = -(-m // bm) * -(-N // bn) * par
                t192 = -(-m // 192) * -(-N // 128) * par
                t128 = -(-m // 128) * -(-N // 128) * par
                t64, t96x64, t128x64 = -(-m // 64) * -(-N // 64) * par, -(-m // 96) * -(-N // 64) * par, -(-m // 128) * -(-N // 64) * par
                if t192 >= 384:
                    # round 6: an encoder GEMM with more than 512 tiles of 192 x 128 (a second, mostly empty round of 2 workgroups per CU) on ONE round of 256 x 256 tiles
                    # (ragged last m-tile); else a wide GEMM on 128 x 128 tiles while those fit two rounds of the 512 slots
                    one_round_256 = epi in (0, 1, 6) and t192 > 512 and K >= 384 and K % 128 == 0 and N % 256 == 0 and -(-m // 256) * (N // 256) <= 256
                    two_rounds_128 = not one_round_256 and epi in (0, 1) and t192 > 512 and t128 <= 1024
                    assert v == (3 if one_round_256 else 1 if two_rounds_128 else 11 if epi == 6 else 8), (D, n, epi, v)
                    assert g == (8 if epi in (0, 1) and v != 1 else 0), (D, n, epi, v, g)
                    continue
                if epi in (0, 1, 6) and t128 > 512:   # round 6: 128 x 128 tiles beyond the 512 resident slots -> the default tile (its tiles fit them)
                    assert v == (11 if epi == 6 else 8) and tiles <= 512 and g == (8 if epi in (0, 1) else 0), (D, n, epi, v, g)
                    continue
                assert v not in (8, 11, 3) and g == (8 if v == 20 and epi in (0, 1) else 0), (D, n, epi, v, g)
                if v == 20 and epi in (0, 1):   # round 6: one round of 192 x 128 tiles where 128 x 128 tiles would need a second, mostly empty one
                    assert K >= 1024 and tiles <= 256 and t128 > 256 and m % 192 == 0
                if epi == 6 and t64 > 512:   # round 6: the ladder of the residual GEMMs beyond the 512 resident 64 x 64 tiles
                    assert v == (41 if t96x64 <= 448 else 15 if t128x64 <= 512 else 20 if t192 <= 256 else 1), (D, n, v)
                else:
                    assert v != 41 and (v != 20 or epi in (0, 1))
                if v == 1:
                    assert tiles >= 256
                if v in (15, 30, 31) or (v == 12 and K < 2048):
                    assert tiles <= slots, (D, n, epi, v, tiles)              # one round: every tile resident at once
                if v == 30:
                    assert tiles <= 256 and -(-m // 32) * -(-N // 64) * par > 256
                if v == 31:
                    assert tiles <= 256
                if two:
                    assert K % 128 == 0
                if v in (15, 41):
                    assert epi == 6 and t64 > 512 and tiles <= slots
                if v == 9:
                    assert -(-m // 64) * -(-N // 64) * par > 512 and K < 2048
    # operating points of round 5 (in situ, profiles/small_batch_r5.txt): (qkv, fc1, proj, fc2)
    enc = lambda D, n: tuple(pick(e, 192 * n, N, K)[0] for e, N, K in ((0, 3 * D, D), (1, 4 * D, D), (6, D, D), (6, D, 4 * D)))
    assert enc(1024, 1) == (30, 30, 31, 31) and enc(1024, 2) == (12, 12, 31, 31) and enc(1024, 4) == (9, 9, 30, 30)      # ViTPose-L
    assert enc(1024, 8) == (20, 20, 12, 12) and enc(1024, 7)[:2] == (20, 20) and enc(1280, 8)[:2] == (20, 1) and enc(1024, 6)[:2] == (9, 20)   # round 6: one round of 192 x 128 tiles
    assert enc(1024, 12) == (1, 8, 41, 41) and enc(1024, 16)[2:] == (15, 15) and enc(1024, 24)[2:] == (20, 20)      # round 6 (calls 18-19): 96 x 64 at 11-14 crops, one round of 192 x 128 at 22-32
    assert enc(1024, 11) == (1, 8, 41, 41) and enc(1024, 17)[1] == 3 and enc(1024, 21)[1:] == (3, 15, 15) and enc(1024, 22)[1] == 8 and enc(1024, 32)[2:] == (20, 20)
    assert enc(1024, 40)[2:] == (1, 1) and enc(1024, 44)[2:] == (11, 11) and enc(1024, 65)[2:] == (3, 3) and enc(1024, 85)[2:] == (3, 3) and enc(1024, 86)[2:] == (11, 11)   # calls 20-21
    assert enc(768, 1) == (31, 30, 31, 31) and enc(768, 4) == (12, 9, 30, 30) and enc(768, 8) == (9, 1, 12, 12)          # ViTPose-B
    assert enc(768, 12)[2:] == (12, 12) and enc(768, 16)[2:] == (41, 41) and enc(768, 20)[2:] == (15, 15) and enc(768, 24)[2:] == (15, 15) and enc(768, 32)[2:] == (20, 20)
    assert enc(768, 15) == (1, 8, 41, 41) and enc(768, 22)[1] == 3 and enc(768, 27)[1] == 3 and enc(768, 29)[1] == 8 and enc(768, 52)[2:] == (1, 1) and enc(768, 60)[2:] == (11, 11)
    assert enc(1280, 1) == (30, 30, 31, 31) and enc(1280, 4)[2:] == (30, 30) and enc(1280, 8)[2:] == (12, 12) and enc(1280, 12)[2:] == (15, 15)   # ViTPose-H
    assert enc(1280, 10)[1:] == (8, 41, 41) and enc(1280, 14)[1] == 3 and enc(1280, 20)[2:] == (20, 20) and enc(1280, 32)[2:] == (1, 1) and enc(1280, 36)[2:] == (11, 11)
    assert enc(768, 86)[2:] == (3, 3) and enc(768, 113)[2:] == (3, 3) and enc(768, 114)[2:] == (11, 11)
    assert enc(384, 1) == (31, 31, 31, 31) and enc(384, 8) == (12, 9, 30, 30) and enc(384, 32)[2:] == (41, 41) and enc(384, 40)[0] == 8   # ViTPose-S
    assert enc(384, 43)[1] == 3 and enc(384, 55)[1] == 3 and enc(384, 57)[0] == 1 and enc(384, 75)[0] == 1 and enc(384, 76)[0] == 8
    assert enc(768, 256) == (8, 8, 11, 11)                                                                                # BASELINE batch: the default tile (before the 8-phase kernel takes over)


def test_padded_encoder_batch_rule_over_every_batch_size():
    """Round 6: the batch the encoder runs for a chunk of n crops (tile_rules.hip pick_run_batch through the host-only tap vp_dbg_run_batch) walked over every batch size of
    every model.  Invariants: never fewer crops than asked, at most 3 more, a multiple of 4 whenever it differs, never below 33 crops, never beyond the workspace limit,
    multiples of 4 untouched; a padded batch always has an 8-phase tile for mlp.fc1 or mlp.fc2 (that is what the padding buys).  And the sizes measured in
    profiles/small_batch_r6.txt (calls 20, 22) keep their choices."""
    lib = capi.load_library()
    tiles = C.c_int32()
    for D in (384, 768, 1024, 1280):
        for n in range(1, 641):
            r = lib.vp_dbg_run_batch(n, D, 1024)
            assert n <= r <= n + 3, (D, n, r)
            if r != n:
                assert n >= 33 and r % 4 == 0 and n % 4 != 0, (D, n, r)
                assert lib.vp_dbg_gemm8_pick(192 * r, 4 * D, 1, 3, C.byref(tiles)) or lib.vp_dbg_gemm8_pick(192 * r, D, 0, 3, C.byref(tiles)), (D, n, r)
            assert lib.vp_dbg_run_batch(n, D, n) == n                        # a handle of max_batch n that is a multiple of 4 has no room to pad
            assert lib.vp_dbg_run_batch(n, D, (n + 3) // 4 * 4) == r         # ... and one whose workspaces are rounded up (vp_create) does
    run = lambda D, n: lib.vp_dbg_run_batch(n, D, 1024)
    assert [run(768, n) for n in (8, 32, 43, 44, 45, 47, 85, 86, 87, 113, 117)] == [8, 32, 44, 44, 48, 48, 85, 88, 88, 113, 117]     # ViTPose-B
    assert [run(1024, n) for n in (21, 33, 35, 41, 65, 67, 69, 83)] == [21, 36, 36, 41, 68, 68, 72, 84]                              # ViTPose-L
    assert [run(1280, n) for n in (31, 53, 66)] == [31, 56, 68] and [run(384, n) for n in (43, 55, 60)] == [44, 56, 60]              # ViTPose-H, -S
    assert run(768, 256) == 256 and run(1280, 128) == 128 and run(1024, 64) == 64                                                     # the BASELINE batches are multiples of 4


def test_gemm8_tile_selection_over_every_batch_size():
    """The rule that picks the 8-phase kernel's tile (tile_rules.hip pick_gemm8_tile, through the host-only tap vp_dbg_gemm8_pick) walked over every
    batch size 1..640 of every model.  Invariants: a picked tile divides the matrix and the tap reports its tile count; the wide GEMMs never get the
    192-column tile; the mask bits switch the 192-row tile off per GEMM kind; a pick is justified -- it fills its rounds (>= 448 tiles, >= 192 with a last
    round >= 80 % full, or one round from 192 tiles) or it needs fewer rounds x tile area than the 2-phase kernel; nothing is picked below 192 tiles;
    with the round-4 extensions off (mask bit 4) the 192-row tile appears only where no 256-row tile qualifies.  And the operating points measured in
    round 4 (profiles/tile_sweep_r4.txt, profiles/gemm8_bm192_r4.txt) keep their kernels."""
    lib = capi.load_library()
    tiles = C.c_int32()
    dims = {16: (256, 256), 17: (256, 192), 18: (192, 256)}

    def pick(M, N, wide, mask=3):
        v = lib.vp_dbg_gemm8_pick(M, N, wide, mask, C.byref(tiles))
        return v, tiles.value

    def fills(t):
        return t >= 448 or (t >= 192 and t / (-(-t // 256) * 256) >= 0.8)

    for D in (384, 768, 1024, 1280):
        for n in range(1, 641):
            M = 192 * n
            for N, wide in ((3 * D, 1), (4 * D, 1), (D, 0)):
                v, t = pick(M, N, wide)
                v3, t3 = pick(M, N, wide, 4 | 3)                                   # round-3 thresholds + the 192-row fallback
                v0, _ = pick(M, N, wide, 4)
                assert v in (0, 16, 17, 18) and v3 in (0, 16, 17, 18) and v0 in (0, 16, 17)
                assert pick(M, N, wide, 0)[0] != 18 and pick(M, N, wide, 2 if not wide else 1)[0] != 18
                t2 = n * -(-N // 128)
                cost2 = 24576 if t2 <= 256 else -(-t2 // 512) * 49152
                for vv, tt in ((v, t), (v3, t3)):
                    if vv:
                        bm, bn = dims[vv]
                        assert M % bm == 0 and N % bn == 0 and tt == (M // bm) * (N // bn) and tt >= 192, (D, n, N, wide, vv, tt)
                        assert not (wide and vv == 17)
                if v:
                    bm, bn = dims[v]
                    cost = -(-t // 256) * bm * bn * (1.08 if v == 18 else 1.0)
                    assert fills(t) or (M >= 7680 and (t <= 256 or cost < 0.95 * cost2)), (D, n, N, wide, v, t)
                if v3:
                    assert fills(t3)
                    if v3 == 18:
                        assert v0 == 0, (D, n, N, wide)                                # only where no 256-row tile qualifies
                    else:
                        assert v3 == v0
                if M < 7680:
                    assert (v, t) == (v3, t3)                                          # the extensions start at 40 crops
    # operating points (BASELINE batches and the ones measured in round 4): (variant, tiles)
    assert pick(49152, 3072, 1) == (16, 2304) and pick(49152, 768, 0) == (17, 768)            # ViTPose-B 256: fc1, fc2
    assert pick(24576, 5120, 1) == (16, 1920) and pick(24576, 1280, 0) == (16, 480)           # ViTPose-H 128
    assert pick(12288, 4096, 1) == (16, 768) and pick(12288, 1024, 0) == (18, 256)            # ViTPose-L 64: fc2 on 192 x 256 (192 tiles of 256 x 256 = 75 %)
    assert pick(12096, 4096, 1) == (18, 1008) and pick(12096, 1024, 0) == (18, 252)           # ViTPose-L 63: 252 tiles = 252 workgroups, one round
    assert pick(16320, 3072, 1) == (18, 1020) and pick(16320, 768, 0) == (18, 255)            # ViTPose-B 85
    assert pick(16896, 768, 0) == (16, 198) and pick(33024, 768, 0) == (16, 387)              # ViTPose-B 88 / 172: fewer rounds than anything else (+7.6 % / +5.2 %)
    assert pick(24576, 768, 0)[0] == 0 and pick(9984, 768, 0)[0] == 0 and pick(7680, 1024, 0)[0] == 0   # ViTPose-B 128 / 52, -L 40: the 2-phase kernel wins in situ
    assert pick(49152, 384, 0)[0] == 0                                                         # ViTPose-S 256 fc2
    assert pick(16128, 768, 0) == (17, 252) and pick(12288, 768, 0) == (17, 192)              # ViTPose-B 84 / 64
    assert pick(12096, 1024, 0, 0)[0] == 0 and pick(12096, 1024, 0, 2)[0] == 0 and pick(12096, 4096, 1, 1)[0] == 0   # the mask bits: 1 = residual, 2 = wide


@pytest.mark.parametrize('n', [0, 1, 7, 64, 513])
@pytest.mark.parametrize('w', [1, 2, 8])
@pytest.mark.parametrize('maxb', [1, 8, 64])
def test_group_sharding_plan(n, w, maxb):
    """vp_group_infer's sharding (vitpose_api.hip group_plan, the function group_run executes) through the host-only tap
    vp_dbg_group_plan: rounds of w * maxb crops, contiguous shards of ceil(nr / w) crops, no crop lost or duplicated, no shard
    above max_batch, trailing devices short or empty -- for uneven tails, more devices than crops and multi-round calls.
    No device needed: the multi-GPU path is checked before 8-GPU hardware shows up."""
    lib = capi.load_library()
    need = lib.vp_dbg_group_plan(n, w, maxb, None, None, 0)
    rounds = -(-n // (w * maxb)) if n else 0
    assert need == rounds * w
    offs = (C.c_int32 * max(need, 1))()
    cnts = (C.c_int32 * max(need, 1))()
    assert lib.vp_dbg_group_plan(n, w, maxb, offs, cnts, need) == need
    covered = []
    for r in range(rounds):
        nr = min(n - r * w * maxb, w * maxb)
        per = -(-nr // w)
        seen_short = False
        for i in range(w):
            off, cnt = offs[r * w + i], cnts[r * w + i]
            assert 0 <= cnt <= min(per, maxb)
            if cnt < per:
                seen_short = True
            else:
                assert not seen_short                  # full shards first, then at most one short one, then empty ones
            if cnt:
                covered.extend(range(off, off + cnt))
                assert off == r * w * maxb + i * per   # the same bounds the one-process-per-GPU host uses
    assert covered == list(range(n))
    # agreement with easy_vitpose_amd.parallel.shard_bounds (RCCL host) inside one round
    from easy_vitpose_amd.parallel import shard_bounds
    if rounds == 1:
        for i in range(w):
            lo, hi = shard_bounds(n, w, i)
            assert (offs[i], cnts[i]) == (min(lo, n), hi - lo) or (cnts[i] == 0 and hi == lo)
    assert lib.vp_dbg_group_plan(-1, w, maxb, None, None, 0) < 0 and lib.vp_dbg_group_plan(n, 0, maxb, None, None, 0) < 0


@pytest.mark.parametrize('n,w,maxb', [(64, 8, 8), (513, 8, 64), (7, 8, 64), (2048, 8, 256), (5, 2, 1), (1, 8, 8)])
def test_group_call_enqueues_every_member_before_it_waits_for_any(n, w, maxb):
    """The two-phase schedule of vp_group_infer (vitpose_api.hip group_rounds, the function group_run executes) with stub members,
    host only: inside every round ALL submissions precede the first wait, each member that got work is waited for exactly once, in
    member order, and the submissions of round r + 1 start only after round r has been collected (the members' two slots are then
    free again).  Phase 1 of the real path contains no wait either: downloads land in pinned staging, uploads are asynchronous from
    pinned caller memory or staged in pieces."""
    lib = capi.load_library()
    need = lib.vp_dbg_group_trace(n, w, maxb, None, 0)
    plan_n = lib.vp_dbg_group_plan(n, w, maxb, None, None, 0)
    cnts = (C.c_int32 * max(plan_n, 1))()
    offs = (C.c_int32 * max(plan_n, 1))()
    lib.vp_dbg_group_plan(n, w, maxb, offs, cnts, plan_n)
    working = sum(1 for e in range(plan_n) if cnts[e] > 0)
    assert need == 2 * working
    tr = (C.c_int32 * max(need, 1))()
    assert lib.vp_dbg_group_trace(n, w, maxb, tr, need) == need
    tr = list(tr)[:need]
    pos = 0
    for r in range(plan_n // w):
        members = [i + 1 for i in range(w) if cnts[r * w + i] > 0]
        k = len(members)
        assert tr[pos:pos + k] == members, f'round {r}: submissions {tr[pos:pos + k]}'           # phase 1: every member, no wait in between
        assert tr[pos + k:pos + 2 * k] == [-m for m in members], f'round {r}: waits {tr[pos + k:pos + 2 * k]}'   # phase 2
        pos += 2 * k
    assert pos == need
    assert lib.vp_dbg_group_trace(-1, w, maxb, None, 0) < 0


def test_bad_config_is_rejected_before_touching_a_device():
    lib = capi.load_library()
    h = C.c_void_p()
    for cfg in [capi.vp_config(770, 12, 12, 17, 0, 0, 4), capi.vp_config(768, 12, 7, 17, 0, 0, 4),
                capi.vp_config(768, 12, 12, 0, 0, 0, 4), capi.vp_config(768, 12, 12, 17, 9, 0, 4)]:
        assert lib.vp_create(C.byref(h), C.byref(cfg)) == capi.VP_ERR_INVALID
        assert capi.last_error()
    assert lib.vp_create(None, None) == capi.VP_ERR_INVALID


def test_null_handles_and_empty_groups_are_rejected_without_a_device():
    """Every entry point validates its handle / pointers before it touches HIP: VP_ERR_INVALID, never a crash."""
    lib = capi.load_library()
    slot = C.c_int32(-1)
    buf = np.zeros(16, np.float32)
    assert lib.vp_infer(None, buf.ctypes.data, 0, 1, None, buf.ctypes.data) == capi.VP_ERR_INVALID
    assert lib.vp_infer_submit(None, buf.ctypes.data, 0, 1, None, buf.ctypes.data, C.byref(slot)) == capi.VP_ERR_INVALID
    assert lib.vp_infer_wait(None, 0) == capi.VP_ERR_INVALID
    assert lib.vp_group_size(None) == 0
    g = C.c_void_p()
    cfg = capi.vp_config(384, 12, 12, 17, 0, 0, 2)
    ids = (C.c_int32 * 1)(0)
    assert lib.vp_group_create(C.byref(g), C.byref(cfg), ids, 0) == capi.VP_ERR_INVALID
    assert lib.vp_group_create(None, C.byref(cfg), ids, 1) == capi.VP_ERR_INVALID
    assert lib.vp_group_infer(None, buf.ctypes.data, 0, 1, None, buf.ctypes.data) == capi.VP_ERR_INVALID
    lib.vp_group_destroy(None)      # a no-op, like vp_destroy(NULL)
    lib.vp_destroy(None)


def test_no_cpu_fallback_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    lib = capi.load_library()
    h = C.c_void_p()
    cfg = capi.vp_config(384, 12, 12, 17, 0, 0, 2)
    assert lib.vp_create(C.byref(h), C.byref(cfg)) == capi.VP_ERR_HIP
    assert 'no CPU fallback' in capi.last_error()
    out = np.zeros((1, 17, 3), np.float32)
    hm = np.zeros((1, 17, 64, 48), np.float32)
    assert lib.vp_decode_only(0, hm.ctypes.data, 1, 17, None, out.ctypes.data) == capi.VP_ERR_HIP
    from easy_vitpose_amd import VitPoseHip
    from easy_vitpose_amd.synth import synthetic_state_dict
    shp = configs.model_shape('s', 'coco')
    with pytest.raises(capi.VpError):
        VitPoseHip(shp, {}, max_batch=1)


def test_missing_extension_raises(monkeypatch):
    monkeypatch.setattr(capi, '_lib', None)
    monkeypatch.setattr(capi, 'LIB_PATH', '/nonexistent/libvitpose_hip.so')
    with pytest.raises(capi.HipExtensionMissing):
        capi.load_library()


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under easy_vitpose_amd/ may reference it."""
    pkg = os.path.join(ROOT, 'easy_vitpose_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f), errors='ignore').read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), f'{f} imports the oracle'


def _build_c_demo(tmp_path):
    import shutil
    import subprocess
    if shutil.which('gcc') is None or not os.path.exists(capi.LIB_PATH):
        pytest.skip('gcc or the built library is missing')
    exe = str(tmp_path / 'c_api_demo')
    libdir = os.path.dirname(capi.LIB_PATH)
    cmd = ['gcc', '-std=c99', '-Wall', '-Werror', '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'examples', 'c_api_demo.c'),
           '-o', exe, '-L' + libdir, '-lvitpose_hip', '-Wl,-rpath,' + libdir, '-lm']
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return exe


def test_c_abi_is_plain_c_and_fails_loudly_without_a_gpu(tmp_path):
    """The header is valid C99 and a plain-C caller links against the library; on a box without a HIP device the
    first entry point returns VP_ERR_HIP with a message (no CPU fallback)."""
    import subprocess
    import torch
    exe = _build_c_demo(tmp_path)
    if torch.cuda.is_available():
        pytest.skip('GPU present: covered by the gpu-marked run of the same program')
    res = subprocess.run([exe], capture_output=True, text=True)
    assert res.returncode == 2 and 'no CPU fallback' in res.stderr


def test_json_wire_format_like_reference_cli(tmp_path):
    """`--save-json` layout of the reference CLI (inference.py:136-141): {"keypoints": [per-frame {id: [[y,x,score]..]}],
    "skeleton": {index: name}} with ndarrays serialised as nested lists."""
    import json
    from easy_vitpose_amd.jsonio import COCO17_JOINTS, frames_to_json, save_json
    kp = np.arange(17 * 3, dtype=np.float32).reshape(17, 3)
    frames = [{0: kp, 1: kp + 1}, {}, {np.int64(7): kp}]
    d = json.loads(frames_to_json(frames, COCO17_JOINTS))
    assert sorted(d) == ['keypoints', 'skeleton'] and len(d['keypoints']) == 3 and d['keypoints'][1] == {}
    assert d['keypoints'][0]['1'][2] == [7.0, 8.0, 9.0] and list(d['keypoints'][2]) == ['7']
    assert d['skeleton']['0'] == 'nose' and len(d['skeleton']) == 17
    save_json(str(tmp_path / 'o.json'), frames)
    assert json.load(open(tmp_path / 'o.json'))['skeleton'] == {}


# ----------------------------------------------------------------------------- tracker (f-4)
@pytest.mark.parametrize('tag', ['sort_age1', 'sort_age3'])
def test_tracker_matches_reference_sort(golden_dir, tag):
    """easy_vitpose_amd.tracker.Sort against the output of the REFERENCE's Sort (sort.py:203-266, run by make_golden.py with a
    textbook KalmanFilter in place of the absent filterpy) on a seeded detection sequence: same boxes, scores and ids per frame."""
    import os
    from cases import tracker_sequence
    from easy_vitpose_amd.tracker import Sort
    z = np.load(os.path.join(golden_dir, f'{tag}.npz'))
    trk = Sort(max_age=int(z['max_age']), min_hits=3, iou_threshold=0.3)
    rows = []
    for i, d in enumerate(tracker_sequence()):
        o = trk.update(d.copy()).reshape(-1, 6)
        rows.append(np.concatenate([np.full((len(o), 1), i, dtype=np.float64), o], 1))
    got = np.concatenate(rows)
    assert got.shape == z['rows'].shape
    assert np.array_equal(got[:, [0, 6]], z['rows'][:, [0, 6]])          # frame index and track id of every reported box
    assert np.abs(got - z['rows']).max() < 1e-9


@pytest.mark.parametrize('step', [2, 3])
def test_tracker_with_skipped_detector_frames_matches_reference(golden_dir, step):
    """yolo_step > 1: the reference builds Sort(max_age=step, min_hits=1) (inference.py:179-184) and feeds it empty detections
    on the frames where the detector is skipped (:235-236).  Same boxes / scores / ids per frame as the reference's Sort --
    with min_hits = 3 there every detector frame after the third would report NOBODY."""
    import os
    from cases import tracker_sequence_step
    from easy_vitpose_amd.tracker import Sort
    z = np.load(os.path.join(golden_dir, f'sort_step{step}.npz'))
    assert int(z['min_hits']) == 1 and int(z['max_age']) == step
    trk, bad = Sort(max_age=step, min_hits=1, iou_threshold=0.3), Sort(max_age=step, min_hits=3, iou_threshold=0.3)
    rows, empty_with_3 = [], 0
    for i, d in enumerate(tracker_sequence_step(step)):
        o = trk.update(d.copy()).reshape(-1, 6)
        rows.append(np.concatenate([np.full((len(o), 1), i, dtype=np.float64), o], 1))
        ob = bad.update(d.copy())
        empty_with_3 += int(len(d) > 0 and i >= 3 and len(ob) == 0)
    got = np.concatenate(rows)
    assert got.shape == z['rows'].shape
    assert np.array_equal(got[:, [0, 6]], z['rows'][:, [0, 6]])
    assert np.abs(got - z['rows']).max() < 1e-9
    assert empty_with_3 > 0          # the failure mode the default must avoid is real on this sequence


def test_vitinference_reset_uses_the_references_tracker_parameters():
    """VitInference.reset(): Sort(max_age=yolo_step, min_hits=3 if yolo_step == 1 else 1, iou_threshold=0.3) -- inference.py:179-184"""
    from easy_vitpose_amd.inference import VitInference
    for step, hits in [(1, 3), (2, 1), (5, 1)]:
        v = VitInference.__new__(VitInference)
        v.is_video, v.single_pose, v.yolo_step, v._tracker_factory = True, False, step, None
        v.reset()
        assert (v.tracker.max_age, v.tracker.min_hits, v.tracker.iou_threshold) == (step, hits, 0.3) and v.frame_counter == 0
    v.single_pose = True
    v.reset()
    assert v.tracker is None


def test_tracker_contract():
    from easy_vitpose_amd.tracker import Sort, iou_matrix
    assert Sort().update(np.empty((0, 5))).shape == (0, 6)
    t = Sort(max_age=2, min_hits=3, iou_threshold=0.3)
    box = np.array([[10., 10., 50., 90., 0.9]])
    a = t.update(box)
    assert a.shape == (1, 6) and a[0, 5] == 1                            # ids start at 1, young videos report at once
    b = t.update(np.empty((0, 5)))                                       # detector skipped: the predicted box comes back
    assert b.shape == (1, 6) and b[0, 5] == 1
    assert abs(iou_matrix(box, box)[0, 0] - 1.0) < 1e-12
    dot = np.array([[5., 5., 5., 5., 0.5]])
    assert iou_matrix(dot, dot)[0, 0] == 0.0                             # zero-area pair: 0, not NaN (the reference raises there)
    far = np.array([[400., 400., 450., 500., 0.8]])
    c = t.update(np.concatenate([box, far]))
    assert sorted(c[:, 5].astype(int).tolist()) == [1, 2]


def test_cli_argument_surface():
    """the CLI keeps the reference's option names (inference.py:148-187) and refuses what is outside the hot path loudly"""
    from easy_vitpose_amd import cli
    with pytest.raises(AssertionError):
        cli.main(['--input', 'x.png', '--synthetic', 's', '--boxes', 'b.json', '--show'])
    with pytest.raises(AssertionError):
        cli.main(['--input', 'x.png', '--synthetic', 's', '--boxes', 'b.json', '--save-json'])     # needs --output-path
    with pytest.raises(AssertionError):
        cli.main(['--input', 'x.png', '--boxes', 'b.json'])                                        # no model


# ----------------------------------------------------------------------------- resize (f-1)
def test_host_resize_matches_independent_c_oracle():
    """cropprep.resize_linear_u8 (numpy, product host code) against oracle/resize_ref.c, an independent scalar-C restatement of
    OpenCV's 8-bit INTER_LINEAR written from the published algorithm: down- and up-scaling, the exact-2x box path, degenerate
    sizes.  Breaks the circle `device == product host code == golden generator's cv2 stub`."""
    from easy_vitpose_amd.cropprep import resize_linear_u8
    from oracle.resize_ref import resize_linear_u8 as ref
    rng = np.random.default_rng(5)
    for h, w in [(256, 192), (512, 384), (300, 100), (100, 300), (257, 193), (1, 1), (64, 47), (63, 48), (480, 640), (333, 250),
                 (10, 7), (128, 96), (1024, 768), (37, 1000), (2, 3), (255, 191)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(resize_linear_u8(img, (192, 256)), ref(img, (192, 256))), (h, w)
    img = rng.integers(0, 256, (90, 70, 3), dtype=np.uint8)
    for dw, dh in [(35, 45), (140, 180), (71, 89), (1, 1)]:
        assert np.array_equal(resize_linear_u8(img, (dw, dh)), ref(img, (dw, dh))), (dw, dh)


def test_bench_clock_probe_degrades_to_none_without_a_gpu(monkeypatch, tmp_path):
    """bench.py's clock / power pass (rocm-smi polled while the step replays) must never break the bench line: without a GPU (this
    container) or without rocm-smi it returns None; with a well-formed rocm-smi answer it reports the mean clock and the MFMA peak
    the chip can sustain at that clock."""
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    r0 = bench.clock_power_under_load(lambda: time.sleep(0.005), lambda: None, seconds=0.8)
    assert r0 is None or isinstance(r0, dict)
    fake = tmp_path / 'rocm-smi'
    fake.write_text('#!/bin/sh\necho \'{"card0": {"sclk clock speed:": "(1800Mhz)", "Current Socket Graphics Package Power (W)": "1300.0"}}\'\n')
    fake.chmod(0o755)
    monkeypatch.setenv('PATH', f'{tmp_path}:{os.environ["PATH"]}')
    r = bench.clock_power_under_load(lambda: time.sleep(0.005), lambda: None, seconds=1.5)
    assert r is not None and r['sclk_mhz'] == 1800 and r['board_power_w'] == 1300
    assert abs(r['mfma_peak_at_sclk_tflops'] - 2500.0 * 1800 / 2400) < 0.1


def test_tools_and_bench_scripts_compile():
    """tools/*.py, bench.py and __graft_entry__.py are the recipes behind profiles/: they must at least parse (they only run on a GPU box)."""
    import ast
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, 'tools', '*.py'))) + [os.path.join(root, 'bench.py'), os.path.join(root, '__graft_entry__.py')]
    assert len(files) > 20
    for f in files:
        ast.parse(open(f).read(), filename=f)


def test_host_e4m3_converter_matches_torch():
    """The fp8 mode's weight packer rounds on the host (csrc/mx8.h vp_host_e4m3): OCP e4m3fn, round to nearest even, subnormals,
    saturating at 448 -- bit for bit torch.float8_e4m3fn on everything inside the representable range (weights are scaled to
    max 448 before the conversion), incl. every tie and every code value itself."""
    import torch
    F8 = torch.float8_e4m3fn
    lib = capi.load_library()
    allc = torch.arange(256, dtype=torch.uint8).view(F8).float().numpy()
    allc = allc[np.isfinite(allc)]
    pos = np.unique(np.abs(allc))
    mids = (pos[:-1] + pos[1:]) / 2                                   # every tie
    rng = np.random.default_rng(3)
    x = np.concatenate([allc, mids, -mids, np.nextafter(mids, 0).astype(np.float32), np.nextafter(mids, 1e9).astype(np.float32),
                        rng.standard_normal(20000).astype(np.float32) * 100, rng.standard_normal(20000).astype(np.float32) * 0.01,
                        np.float32([0.0, -0.0, 448.0, -448.0, 2.0 ** -9, 2.0 ** -10, 2.0 ** -11, 1e-30, 447.9, 455.9])]).astype(np.float32)
    x = x[np.abs(x) <= 448.0]
    got = np.empty(x.size, np.uint8)
    assert lib.vp_dbg_host_e4m3(x.ctypes.data, got.ctypes.data, x.size) == 0
    ref = torch.from_numpy(x).to(F8).view(torch.uint8).numpy()
    bad = got != ref
    zero = (x == 0) | ((got & 0x7f) == 0) & ((ref & 0x7f) == 0)       # +-0 may differ in sign for values that round to zero
    assert not (bad & ~zero).any(), f'{(bad & ~zero).sum()} codes differ, e.g. x={x[bad & ~zero][:5]} got={got[bad & ~zero][:5]} ref={ref[bad & ~zero][:5]}'
    # saturation beyond the range (torch would give NaN there; the packer never produces such inputs, the converter clamps)
    big = np.float32([460.0, 1e6, -1e6])
    g2 = np.empty(3, np.uint8)
    lib.vp_dbg_host_e4m3(big.ctypes.data, g2.ctypes.data, 3)
    assert g2.tolist() == [0x7e, 0x7e, 0xfe]
