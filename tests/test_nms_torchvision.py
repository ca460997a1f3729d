"""Pin of the ONE third-party boundary of the pseudo-label filter: ``torchvision.ops.nms`` (reference call site
utils/general.py:976; SURVEY.md 8c "parity unpinned": torchvision is not installed in the build container and the reference
holds no golden output for it, so `oracle.nms.nms` is a restatement of the published CPU kernel and the NMS goldens were
generated with that restatement standing in for torchvision -- VERDICT r03 weak 3, "circular").

Wherever a real torchvision is importable (``pytest.importorskip``), these tests close the circle:
  * `oracle.nms.nms`          == torchvision.ops.nms   (keep indices, bit-exact) on the golden candidate sets incl. the tie cases
  * `oracle.nms.non_max_suppression_ssod` re-run with torchvision.ops.nms in place of the restatement gives the same rows / indices
  * the HIP filter (et_nms_ssod) == that torchvision-backed run (``-m gpu``)
Skipped (and reported as skipped, not passed) when torchvision is absent; tools/gpu_round.sh records which it was on the GPU box.
"""
import numpy as np
import pytest
import torch

from oracle import nms as o_nms
from tests.conftest import golden

tv_ops = pytest.importorskip("torchvision.ops", reason="torchvision is not installed: the torchvision.ops.nms pin cannot run here")


def _tv_nms(boxes, scores, iou_thres):
    k = tv_ops.nms(torch.from_numpy(np.ascontiguousarray(boxes, np.float32)), torch.from_numpy(np.ascontiguousarray(scores, np.float32)),
                   float(iou_thres))
    return k.numpy().astype(np.int64)


def _candidates(pred, ct):
    """the (boxes + class offset, scores) matrices the reference hands to torchvision.ops.nms, per image (general.py:921-976)"""
    out = []
    nc = pred.shape[2] - 5
    for x in np.asarray(pred, np.float32):
        x = x[x[:, 4] > np.float32(ct)].copy()
        if not x.shape[0]:
            continue
        x[:, 5:5 + nc] *= x[:, 4:5]
        box = o_nms.xywh2xyxy(x[:, :4])
        j = x[:, 5:5 + nc].argmax(1)[:, None]
        conf = np.take_along_axis(x[:, 5:5 + nc], j, 1)
        sel = conf.reshape(-1) > np.float32(ct)
        if not sel.any():
            continue
        out.append((box[sel] + j[sel].astype(np.float32) * np.float32(o_nms.MAX_WH), conf[sel].reshape(-1)))
    return out


def _tie_heavy(seed, B=2, A=900, nc=6):
    rng = np.random.default_rng(seed)
    pred = np.zeros((B, A, 5 + nc), np.float32)
    centers = rng.uniform(80, 560, (B, 7, 2)).astype(np.float32)
    idx = rng.integers(0, 7, (B, A))
    pred[..., 0:2] = np.take_along_axis(centers, idx[..., None].repeat(2, 2), 1) + rng.normal(0, 4, (B, A, 2))
    pred[..., 2:4] = 60 + rng.normal(0, 8, (B, A, 2))
    pred[..., 4] = np.round(rng.uniform(0.2, 1, (B, A)), 1)            # many exactly equal scores
    pred[..., 5:] = np.round(rng.uniform(0.2, 1, (B, A, nc)), 1)
    pred[:, ::7] = pred[:, 1::7][:, :pred[:, ::7].shape[1]]            # duplicated rows: equal score AND equal box
    return pred


CASES = [("golden", c) for c in ("a", "b", "c")] + [("ties", s) for s in (0, 1, 2)]


def _case(kind, key):
    if kind == "golden":
        g = golden("nms")
        return g[f"{key}_pred"], float(g[f"{key}_thr"][0]), float(g[f"{key}_thr"][1])
    return _tie_heavy(key), 0.1, 0.65


@pytest.mark.parametrize("kind,key", CASES)
def test_restated_nms_equals_torchvision(kind, key):
    pred, ct, it = _case(kind, key)
    cands = _candidates(pred, ct)
    assert cands
    for boxes, scores in cands:
        assert np.array_equal(o_nms.nms(boxes, scores, it), _tv_nms(boxes, scores, it))


@pytest.mark.parametrize("kind,key", CASES)
def test_oracle_filter_is_unchanged_with_the_real_torchvision(kind, key, monkeypatch):
    pred, ct, it = _case(kind, key)
    want, wkeep = o_nms.non_max_suppression_ssod(pred, ct, it)
    monkeypatch.setattr(o_nms, "nms", _tv_nms)
    got, gkeep = o_nms.non_max_suppression_ssod(pred, ct, it)
    for a, b, ka, kb in zip(want, got, wkeep, gkeep):
        assert np.array_equal(a, b) and np.array_equal(ka, kb)


@pytest.mark.parametrize("kind,key", CASES)
def test_hip_filter_equals_the_torchvision_backed_reference_path(hip, kind, key, monkeypatch):
    from efficientteacher_amd.utils.general import nms_ssod_padded
    pred, ct, it = _case(kind, key)
    monkeypatch.setattr(o_nms, "nms", _tv_nms)
    ref, rkeep = o_nms.non_max_suppression_ssod(pred, ct, it)
    dets, counts, keep, _ = nms_ssod_padded(hip.t(pred), ct, it)
    dets, counts, keep = dets.cpu().numpy(), counts.cpu().numpy(), keep.cpu().numpy()
    for i in range(pred.shape[0]):
        assert counts[i] == ref[i].shape[0]
        assert np.array_equal(dets[i, :counts[i]], ref[i]) and np.array_equal(keep[i, :counts[i]], rkeep[i])
