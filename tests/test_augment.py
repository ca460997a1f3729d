"""On-device strong view (csrc/augment.hip).  cv2 is not installed in the build image and the reference has no
image goldens: the pixel arithmetic is PARITY-UNPINNED.  What is tested: the invariants every correct implementation has
(identity, integer shifts, flips, border value, cutout overwrite order, LUT identity), agreement with an independent numpy
restatement of the same published OpenCV algorithms (oracle-style, written here in the test), and the host-side recipe
(matrix composition / inversion / LUT formulas) against numpy."""
import types

import numpy as np
import pytest
import torch


def _ident(B):
    minv = np.tile(np.array([1., 0, 0, 0, 1, 0]), (B, 1))
    return minv, np.zeros((B, 32, 7), np.int32), np.zeros((B, 3), np.int32)


def _run(hip, weak, minv, lut, cuts, flags):
    from efficientteacher_amd import ops
    t = lambda a: None if a is None else hip.t(np.ascontiguousarray(a))
    return ops.strong_view_u8(hip.t(weak), t(minv), t(lut), t(cuts), t(flags)).cpu().numpy()


def _np_warp(img, minv, border=114):
    """cv2.warpAffine INTER_LINEAR / BORDER_CONSTANT restated with numpy integers (AB_BITS 10, INTER_BITS 5, 15-bit weights)"""
    C, H, W = img.shape
    ys, xs = np.mgrid[0:H, 0:W]
    rnd = lambda v: np.rint(v).astype(np.int64)
    X0 = rnd((minv[1] * ys + minv[2]) * 1024.0) + 16
    Y0 = rnd((minv[4] * ys + minv[5]) * 1024.0) + 16
    X = (X0 + rnd(minv[0] * xs * 1024.0)) >> 5
    Y = (Y0 + rnd(minv[3] * xs * 1024.0)) >> 5
    sx, sy, fx, fy = X >> 5, Y >> 5, X & 31, Y & 31
    out = np.zeros_like(img)
    pad = np.full((C, H + 2, W + 2), border, np.int64)

    def at(c, yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        return np.where(ok, img[c][np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64), border)
    for c in range(C):
        v = ((32 - fx) * (32 - fy) * 32 * at(c, sy, sx) + fx * (32 - fy) * 32 * at(c, sy, sx + 1)
             + (32 - fx) * fy * 32 * at(c, sy + 1, sx) + fx * fy * 32 * at(c, sy + 1, sx + 1))
        out[c] = ((v + (1 << 14)) >> 15).astype(np.uint8)
    return out


def test_identity_shift_flip_border(hip):
    rng = np.random.default_rng(0)
    B, H, W = 3, 20, 28
    weak = rng.integers(0, 256, (B, 3, H, W), dtype=np.uint8)
    minv, cuts, flags = _ident(B)
    assert np.array_equal(_run(hip, weak, minv, None, cuts, flags), weak)
    # image 1: content moved by (+3, -2) pixels -> dst(x, y) = src(x - 3, y + 2); uncovered pixels take the border value 114
    minv[1] = [1, 0, -3, 0, 1, 2]
    flags[2] = [0, 1, 1]                                   # image 2: flipped both ways
    out = _run(hip, weak, minv, None, cuts, flags)
    assert np.array_equal(out[0], weak[0])
    exp = np.full_like(weak[1], 114)
    exp[:, :H - 2, 3:] = weak[1][:, 2:, :W - 3]
    assert np.array_equal(out[1], exp)
    assert np.array_equal(out[2], weak[2][:, ::-1, ::-1])


@pytest.mark.parametrize("seed", [1, 2])
def test_warp_matches_the_restated_fixed_point_algorithm(hip, seed):
    """rotation + scale + shear + sub-pixel translation, each image its own matrix"""
    from efficientteacher_amd.utils.augment import affine_matrix, invert_affine
    import random
    rng = np.random.default_rng(seed)
    B, H, W = 2, 40, 56
    weak = rng.integers(0, 256, (B, 3, H, W), dtype=np.uint8)
    minv, cuts, flags = _ident(B)
    pr = random.Random(seed)
    for i in range(B):
        M, _ = affine_matrix(H, W, 10.0, 0.1, 0.5, 2.0, pr)
        minv[i] = invert_affine(M)
    out = _run(hip, weak, minv, None, cuts, flags)
    for i in range(B):
        assert np.array_equal(out[i], _np_warp(weak[i], minv[i]))


def test_cutouts_lut_and_order(hip):
    rng = np.random.default_rng(3)
    B, H, W = 2, 16, 24
    weak = rng.integers(0, 256, (B, 3, H, W), dtype=np.uint8)
    minv, cuts, flags = _ident(B)
    cuts[0, 0] = [2, 3, 10, 9, 10, 20, 30]
    cuts[0, 1] = [6, 5, 14, 12, 200, 100, 50]              # overlaps the first: the later rectangle wins
    flags[0, 0] = 2
    ident_lut = np.tile(np.arange(256, dtype=np.uint8), (B, 3, 1))
    out = _run(hip, weak, minv, ident_lut, cuts, flags)
    exp = weak[0].copy()
    exp[:, 3:9, 2:10] = np.array([10, 20, 30], np.uint8)[:, None, None]
    exp[:, 5:12, 6:14] = np.array([200, 100, 50], np.uint8)[:, None, None]
    inside = np.zeros((H, W), bool); inside[3:9, 2:10] = True; inside[5:12, 6:14] = True
    assert np.array_equal(out[0][:, inside], exp[:, inside])
    # identity LUTs: RGB -> HSV -> RGB in 8 bits is lossy by construction (H has 180 steps), but never by more than a few levels,
    # and exactly the identity on grey pixels (S = 0)
    d = np.abs(out[1].astype(int) - weak[1].astype(int))
    assert d.max() <= 6, d.max()
    grey = np.repeat(rng.integers(0, 256, (1, 1, H, W), dtype=np.uint8), 3, 1)
    og = _run(hip, grey, minv[:1], ident_lut[:1], cuts[:1], np.zeros((1, 3), np.int32))
    assert np.array_equal(og, grey)
    # value LUT halves the brightness of a grey image
    lut = ident_lut[:1].copy()
    lut[0, 2] = (np.arange(256) * 0.5).astype(np.uint8)
    assert np.array_equal(_run(hip, grey, minv[:1], lut, cuts[:1], np.zeros((1, 3), np.int32)), (grey * 0.5).astype(np.uint8))


def test_host_recipe_against_numpy():
    from efficientteacher_amd.utils.augment import StrongViewGenerator, affine_matrix, hsv_luts, invert_affine
    import random
    M, s = affine_matrix(64, 96, 5.0, 0.1, 0.5, 2.0, random.Random(4))
    inv = np.array(invert_affine(M)).reshape(2, 3)
    full = np.linalg.inv(M)
    assert np.allclose(inv, full[:2], rtol=1e-12, atol=1e-9) and 0.5 <= s <= 1.5
    lut = hsv_luts(0.015, 0.7, 0.4, np.random.RandomState(5))
    assert lut.shape == (3, 256) and lut.dtype == np.uint8 and lut[0].max() < 180 and (np.diff(lut[1].astype(int)) >= 0).all()
    hyp = types.SimpleNamespace(degrees=0.0, translate=0.1, scale=0.5, shear=0.0, hsv_h=0.015, hsv_s=0.7, hsv_v=0.4, cutout=1.0,
                                flipud=0.0, fliplr=0.5)
    g = StrongViewGenerator(hyp, seed=6)
    minv, lut, cuts, flags, M_s = g.sample(4, 64, 64)
    assert M_s.shape == (4, 13) and list(M_s[:, 0]) == [0, 1, 2, 3] and (M_s[:, 11] == 0).all()
    for i in range(4):
        assert np.allclose(np.array(minv[i]).reshape(2, 3), np.linalg.inv(M_s[i, 1:10].reshape(3, 3))[:2], atol=1e-9)
        assert flags[i, 0] in (0, 31, 32) and (cuts[i, :flags[i, 0], 2] > cuts[i, :flags[i, 0], 0]).all()


def test_generator_feeds_the_pseudo_label_transform(hip):
    """strong view + M_s from the generator: a box drawn on the weak view lands on the same content in the strong view"""
    from efficientteacher_amd import ops
    from efficientteacher_amd.utils.augment import StrongViewGenerator
    hyp = types.SimpleNamespace(degrees=0.0, translate=0.05, scale=0.2, shear=0.0, hsv_h=0.0, hsv_s=0.0, hsv_v=0.0, cutout=0.0,
                                flipud=0.0, fliplr=1.0)
    B, H, W = 2, 64, 64
    weak = np.full((B, 3, H, W), 30, np.uint8)
    weak[:, :, 20:36, 10:30] = 220                                     # a bright rectangle: x 10..30, y 20..36
    g = StrongViewGenerator(hyp, seed=9)
    strong, M_s = g(hip.t(weak))
    dets = torch.zeros((B, 300, 8), device=hip.device)
    dets[:, 0, :4] = torch.tensor([10., 20., 30., 36.], device=hip.device)
    dets[:, 0, 4:] = torch.tensor([0.9, 1.0, 0.95, 0.95], device=hip.device)
    counts = torch.ones(B, dtype=torch.int32, device=hip.device)
    t9, valid = ops.pseudo_label_transform(dets, counts, M_s, W, H)
    t9 = t9[valid.bool()].cpu().numpy()
    s = strong.cpu().numpy()
    assert t9.shape[0] == B
    for row in t9:
        i = int(row[0])
        x0, x1 = (row[2] - row[4] / 2) * W, (row[2] + row[4] / 2) * W
        y0, y1 = (row[3] - row[5] / 2) * H, (row[3] + row[5] / 2) * H
        box = s[i, 0, int(np.ceil(y0)) + 1:int(y1) - 1, int(np.ceil(x0)) + 1:int(x1) - 1]
        assert box.size > 20 and (box > 200).all(), (row, box.min())
        assert (s[i, 0] > 200).sum() <= 1.5 * (x1 - x0) * (y1 - y0)       # and the bright area is not much larger than the box
