"""TEST INFRASTRUCTURE -- CPU restatement of the anchor assigner.

Restates ``YOLOAnchorAssigner.build_targets``
(models/assigner/yolo_anchor_assigner.py:319-372) and ``build_uc_targets_aug``
(:640-696) in numpy fp32/int64, written as an explicit per-candidate loop nest so
that the *output order*  is visible:

    level -> offset block [centre, j(left), k(up), l(right), m(down)]
          -> anchor (0..na-1) -> target (input order)

``targets`` rows are [img, cls, x, y, w, h] (+ [score] for the uc variant),
normalised.  Returns per level: tcls int64 (n,), tbox fp32 (n,4) [dx,dy,gw,gh],
indices (b,a,gj,gi) int64, anch fp32 (n,2) and, for uc, tscore fp32 (n,).
"""
import numpy as np

F32 = np.float32
OFF = np.array([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1]], F32) * F32(0.5)  # :328-332


def build_targets(shapes, anchors, targets, anchor_t=4.0, with_score=False):
    """shapes: list of (ny, nx) per level; anchors: (nl, na, 2) stride-normalised."""
    targets = np.asarray(targets, F32)
    ncol = 7 if with_score else 6
    targets = targets[:, :ncol].reshape(-1, ncol)
    anchors = np.asarray(anchors, F32)
    nl, na = anchors.shape[:2]
    nt = targets.shape[0]
    at = F32(anchor_t)
    out = []
    for i in range(nl):
        ny, nx = shapes[i]
        gain = np.ones(ncol, F32)
        gain[2:6] = [nx, ny, nx, ny]                                 # :337
        t = targets * gain                                           # :340  (nt, ncol)
        rows_b, rows_a, rows_gj, rows_gi, rows_c = [], [], [], [], []
        rows_box, rows_an, rows_sc = [], [], []
        if nt:
            # :343-344  keep[a, t]
            with np.errstate(divide="ignore", invalid="ignore"):
                r = t[None, :, 4:6] / anchors[i][:, None, :]
                keep = np.maximum(r, F32(1) / r).max(2) < at        # (na, nt)
            gxy = t[:, 2:4]
            gxi = np.array([nx, ny], F32) - gxy                      # :349
            mj = (np.remainder(gxy[:, 0], F32(1)) < F32(0.5)) & (gxy[:, 0] > F32(1))
            mk = (np.remainder(gxy[:, 1], F32(1)) < F32(0.5)) & (gxy[:, 1] > F32(1))
            ml = (np.remainder(gxi[:, 0], F32(1)) < F32(0.5)) & (gxi[:, 0] > F32(1))
            mm = (np.remainder(gxi[:, 1], F32(1)) < F32(0.5)) & (gxi[:, 1] > F32(1))
            masks = [np.ones(nt, bool), mj, mk, ml, mm]              # :352
            for o in range(5):
                for a in range(na):
                    for k in range(nt):
                        if not (keep[a, k] and masks[o][k]):
                            continue
                        gx, gy = t[k, 2], t[k, 3]
                        gi = np.int64(np.trunc(gx - OFF[o, 0]))      # :362 .long()
                        gj = np.int64(np.trunc(gy - OFF[o, 1]))
                        gi = min(max(gi, 0), nx - 1)                 # :367 clamp_ (in place,
                        gj = min(max(gj, 0), ny - 1)                 #  before tbox at :368)
                        rows_b.append(np.int64(np.trunc(t[k, 0])))
                        rows_c.append(np.int64(np.trunc(t[k, 1])))
                        rows_a.append(a); rows_gj.append(gj); rows_gi.append(gi)
                        rows_box.append([gx - F32(gi), gy - F32(gj), t[k, 4], t[k, 5]])
                        rows_an.append(anchors[i][a])
                        if with_score:
                            rows_sc.append(t[k, 6])
        res = dict(
            tcls=np.asarray(rows_c, np.int64),
            tbox=np.asarray(rows_box, F32).reshape(-1, 4),
            b=np.asarray(rows_b, np.int64), a=np.asarray(rows_a, np.int64),
            gj=np.asarray(rows_gj, np.int64), gi=np.asarray(rows_gi, np.int64),
            anch=np.asarray(rows_an, F32).reshape(-1, 2))
        if with_score:
            res["tscore"] = np.asarray(rows_sc, F32)
        out.append(res)
    return out
