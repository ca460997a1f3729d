"""Strong-view generation for the unlabeled stream on the device (reference utils/datasets_ssod.py:520-570).

The reference's data-loader workers produce BOTH views with cv2 on the host.  ``StrongViewGenerator`` keeps their random
recipe on the host -- a few numbers per image -- and does the per-pixel work in one kernel (csrc/augment.hip):

    M = T @ S @ R @ C            random_perspective_with_M   (datasets_ssod.py:902-945; perspective = 0 in every recipe)
    hue / sat / val gains        augment_hsv                 (augmentations.py:48-61)
    cutout rectangles + colours  cutout                      (augmentations.py:382-398; label filtering stays with the labels)
    flipud / fliplr                                          (datasets_ssod.py:552-563)

and returns the strong batch together with the ``M_s`` rows ``[i, M(9), s, ud, lr]`` the pseudo-label transform consumes
(utils/self_supervised_utils.py).  PARITY UNPINNED for the pixels (cv2 is not installed here; see csrc/augment.hip): the
matrices, LUTs and rectangles follow the reference's formulas exactly, given the same random numbers.
"""
import math
import random

import numpy as np
import torch

from .. import ops

MAX_CUT = 32


def affine_matrix(h, w, degrees, translate, scale, shear, rng=random):
    """(M (3,3), s): datasets_ssod.py:909-938 with perspective = 0 and no border"""
    C = np.eye(3)
    C[0, 2] = -w / 2
    C[1, 2] = -h / 2
    rng.uniform(0, 0); rng.uniform(0, 0)                       # the two perspective draws of the reference's stream
    a = rng.uniform(-degrees, degrees)
    s = rng.uniform(1 - scale, 1 + scale)
    R = np.eye(3)
    ar = math.radians(a)                                       # cv2.getRotationMatrix2D(angle=a, center=(0,0), scale=s)
    al, be = s * math.cos(ar), s * math.sin(ar)
    R[:2] = [[al, be, 0.0], [-be, al, 0.0]]
    S = np.eye(3)
    S[0, 1] = math.tan(rng.uniform(-shear, shear) * math.pi / 180)
    S[1, 0] = math.tan(rng.uniform(-shear, shear) * math.pi / 180)
    T = np.eye(3)
    T[0, 2] = rng.uniform(0.5 - translate, 0.5 + translate) * w
    T[1, 2] = rng.uniform(0.5 - translate, 0.5 + translate) * h
    return T @ S @ R @ C, s


def invert_affine(M):
    """cv2.warpAffine's inversion of the 2x3 forward map (imgwarp.cpp: invertAffineTransform semantics), fp64"""
    m0, m1, m2, m3, m4, m5 = (float(M[0, 0]), float(M[0, 1]), float(M[0, 2]), float(M[1, 0]), float(M[1, 1]), float(M[1, 2]))
    D = m0 * m4 - m1 * m3
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m4 * D, m0 * D
    i0, i1, i3, i4 = A11, m1 * -D, m3 * -D, A22
    return [i0, i1, -i0 * m2 - i1 * m5, i3, i4, -i3 * m2 - i4 * m5]


def hsv_luts(hgain, sgain, vgain, rng=np.random):
    """augmentations.py:51-58"""
    r = rng.uniform(-1, 1, 3) * [hgain, sgain, vgain] + 1
    x = np.arange(0, 256, dtype=r.dtype)
    return np.stack((((x * r[0]) % 180).astype(np.uint8), np.clip(x * r[1], 0, 255).astype(np.uint8),
                     np.clip(x * r[2], 0, 255).astype(np.uint8)))


def cutout_rects(h, w, rng=random):
    """augmentations.py:386-398 -> rows (x0, y0, x1, y1, r, g, b); the reference assigns [c0, c1, c2] to a BGR image"""
    rows = []
    for s in [0.5] * 1 + [0.25] * 2 + [0.125] * 4 + [0.0625] * 8 + [0.03125] * 16:
        mask_h = rng.randint(1, int(h * s))
        mask_w = rng.randint(1, int(w * s))
        xmin = max(0, rng.randint(0, w) - mask_w // 2)
        ymin = max(0, rng.randint(0, h) - mask_h // 2)
        xmax, ymax = min(w, xmin + mask_w), min(h, ymin + mask_h)
        c = [rng.randint(64, 191) for _ in range(3)]
        rows.append([xmin, ymin, xmax, ymax, c[2], c[1], c[0]])
    return rows[:MAX_CUT]


class StrongViewGenerator:
    def __init__(self, hyp, seed=None):
        """hyp: the reference's cfg.hyp node (degrees, translate, scale, shear, hsv_h, hsv_s, hsv_v, cutout, flipud, fliplr)"""
        self.hyp = hyp
        self.py_rng = random.Random(seed)
        self.np_rng = np.random.RandomState(seed)

    def sample(self, B, H, W):
        """host side: (minv (B,6) f64, lut (B,3,256) u8, cutouts (B,32,7) i32, flags (B,3) i32, M_s (B,13) f64)"""
        hyp = self.hyp
        minv = np.zeros((B, 6)); lut = np.zeros((B, 3, 256), np.uint8)
        cuts = np.zeros((B, MAX_CUT, 7), np.int32); flags = np.zeros((B, 3), np.int32); M_s = np.zeros((B, 13))
        for i in range(B):
            M, s = affine_matrix(H, W, hyp.degrees, hyp.translate, hyp.scale, hyp.shear, self.py_rng)
            minv[i] = invert_affine(M)
            lut[i] = hsv_luts(hyp.hsv_h, hyp.hsv_s, hyp.hsv_v, self.np_rng)
            if self.py_rng.random() < float(getattr(hyp, "cutout", 0.0)) and self.py_rng.random() < 0.5:     # :540, :384
                rows = cutout_rects(H, W, self.py_rng)
                cuts[i, :len(rows)] = rows
                flags[i, 0] = len(rows)
            flags[i, 1] = int(self.py_rng.random() < hyp.flipud)
            flags[i, 2] = int(self.py_rng.random() < hyp.fliplr)
            M_s[i] = [i, *M.reshape(-1), s, flags[i, 1], flags[i, 2]]
        return minv, lut, cuts, flags, M_s

    def __call__(self, weak_u8):
        """weak (B,3,H,W) uint8 on the device -> (strong (B,3,H,W) uint8, M_s (B,13) fp64 on the device)"""
        B, _, H, W = weak_u8.shape
        minv, lut, cuts, flags, M_s = self.sample(B, H, W)
        dev = weak_u8.device
        t = lambda a: torch.from_numpy(a).to(dev, non_blocking=True)
        strong = ops.strong_view_u8(weak_u8.contiguous(), t(minv), t(lut), t(cuts), t(flags))
        return strong, t(M_s)


# ---- 4-image mosaic (load_mosaic_with_M, utils/datasets_ssod.py:732-792) ---------------------------------------------------
def mosaic_layout(s, yc, xc, shapes):
    """Placement of the four tiles around the centre (xc, yc) of the 2s x 2s canvas (datasets_ssod.py:746-762).
    shapes: [(h, w)] * 4 in tile order (top left, top right, bottom left, bottom right).
    -> rows (x1a, y1a, x2a, y2a, x1b, y1b, padw, padh): canvas rectangle, its origin inside the image, label offsets"""
    rows = []
    for i, (h, w) in enumerate(shapes):
        if i == 0:
            x1a, y1a, x2a, y2a = max(xc - w, 0), max(yc - h, 0), xc, yc
            x1b, y1b = w - (x2a - x1a), h - (y2a - y1a)
        elif i == 1:
            x1a, y1a, x2a, y2a = xc, max(yc - h, 0), min(xc + w, s * 2), yc
            x1b, y1b = 0, h - (y2a - y1a)
        elif i == 2:
            x1a, y1a, x2a, y2a = max(xc - w, 0), yc, xc, min(s * 2, yc + h)
            x1b, y1b = w - (x2a - x1a), 0
        else:
            x1a, y1a, x2a, y2a = xc, yc, min(xc + w, s * 2), min(s * 2, yc + h)
            x1b, y1b = 0, 0
        rows.append((x1a, y1a, x2a, y2a, x1b, y1b, x1a - x1b, y1a - y1b))
    return rows


def mosaic_labels(s, layout, shapes, labels):
    """labels: four (n, 5) arrays [cls, x, y, w, h] normalised to their image -> (N, 5) [cls, x1, y1, x2, y2] in the frame of the
    s x s mosaic: xywhn2xyxy with w / 2, h / 2, padw / 2, padh / 2 (datasets_ssod.py:768: the canvas is halved by the resize that
    follows), then the reference's clip to [0, 2s] (:775-776 -- it clips to the CANVAS size although the coordinates are already
    halved: kept as it is)"""
    out = []
    for (x1a, y1a, x2a, y2a, x1b, y1b, padw, padh), (h, w), lb in zip(layout, shapes, labels):
        lb = np.asarray(lb, dtype=np.float64).reshape(-1, 5).copy()
        if lb.size:
            x = lb[:, 1:].copy()
            lb[:, 1] = (w / 2) * (x[:, 0] - x[:, 2] / 2) + padw / 2
            lb[:, 2] = (h / 2) * (x[:, 1] - x[:, 3] / 2) + padh / 2
            lb[:, 3] = (w / 2) * (x[:, 0] + x[:, 2] / 2) + padw / 2
            lb[:, 4] = (h / 2) * (x[:, 1] + x[:, 3] / 2) + padh / 2
        out.append(lb)
    out = np.concatenate(out, 0)
    np.clip(out[:, 1:], 0, 2 * s, out=out[:, 1:])
    return out


class MosaicGenerator:
    """The 4-image mosaic of the reference's SSOD loader (load_mosaic_with_M) with the per-pixel work on the device: the host draws
    the centre and the three partner images exactly as the reference does (same calls on the same `random` stream: two
    `random.uniform` for (yc, xc), `random.choices(indices, k=3)`, `random.shuffle`) and lays the tiles out; ONE kernel
    (et_mosaic4_u8, csrc/augment.hip) then writes the s x s mosaic -- the 2s x 2s canvas of border value 114 is never materialised:
    every output pixel averages its 2 x 2 canvas pixels, which is what cv2.resize(img4, (s, s)) computes at this exact 2:1 ratio
    (OpenCV switches INTER_LINEAR to its fast INTER_AREA path there: (a + b + c + d + 2) >> 2).  The result is the WEAK view
    (`img4_ori`); the strong view and M_s come from StrongViewGenerator on it, as random_perspective_with_M follows in the reference.
    PARITY: placement, label arithmetic and the consumption of the random stream are pinned on the live reference
    (tests/golden/mosaic.npz); the 2:1 resampling restates OpenCV's published algorithm and is unpinned (no cv2 here)."""

    def __init__(self, img_size, seed=None):
        self.s = int(img_size)
        self.border = [-self.s // 2, -self.s // 2]             # datasets_ssod.py:260
        self.rng = random.Random(seed)

    def sample(self, index, indices):
        """-> (yc, xc, the four image indices in tile order)"""
        s = self.s
        yc, xc = [int(self.rng.uniform(-x, 2 * s + x)) for x in self.border]
        four = [index] + self.rng.choices(indices, k=3)
        self.rng.shuffle(four)
        return yc, xc, four

    def __call__(self, images, labels, index, indices=None):
        """images: dict / list index -> uint8 device tensor (3, h, w) with max(h, w) <= img_size (what load_image returns, as RGB
        planes); labels: index -> (n, 5) normalised [cls, x, y, w, h].  -> (mosaic (3, s, s) uint8 on the device, labels (N, 5) xyxy)"""
        indices = list(range(len(images))) if indices is None else list(indices)
        yc, xc, four = self.sample(index, indices)
        tiles = [images[i] for i in four]
        shapes = [(int(t.shape[1]), int(t.shape[2])) for t in tiles]
        layout = mosaic_layout(self.s, yc, xc, shapes)
        out = ops.mosaic4_u8([tiles], [layout], self.s)[0]
        return out, mosaic_labels(self.s, layout, shapes, [labels[i] for i in four])
