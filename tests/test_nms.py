"""non_max_suppression_ssod: HIP path vs oracle (bit-exact) and vs the reference's golden output."""
import numpy as np
import pytest
import torch

from oracle import nms as o_nms
from tests.conftest import golden


def _run(hip, pred, ct, it, max_det=300):
    from efficientteacher_amd.utils.general import nms_ssod_padded
    dets, counts, keep, ncand = nms_ssod_padded(hip.t(pred), ct, it, max_det=max_det)
    return dets.cpu().numpy(), counts.cpu().numpy(), keep.cpu().numpy(), ncand.cpu().numpy()


@pytest.mark.parametrize("case", ["a", "b", "c", "empty", "tie06"])
def test_nms_golden(hip, case):
    g = golden("nms")
    pred, (ct, it) = g[f"{case}_pred"], g[f"{case}_thr"]
    dets, counts, keep, _ = _run(hip, pred, float(ct), float(it))
    assert np.array_equal(counts, g[f"{case}_counts"])
    got = np.concatenate([dets[i, :c] for i, c in enumerate(counts)], 0)
    assert np.array_equal(got, g[f"{case}_dets"])                       # bit exact rows
    gk = np.concatenate([keep[i, :c] for i, c in enumerate(counts)], 0)
    assert np.array_equal(gk, g[f"{case}_keep"])                        # bit exact indices
    for i, c in enumerate(counts):
        assert (keep[i, c:] == -1).all() and (dets[i, c:] == 0).all()


@pytest.mark.parametrize("seed,B,A,nc", [(0, 2, 700, 80), (1, 1, 64, 3), (2, 3, 333, 1), (3, 2, 1300, 20)])
def test_nms_vs_oracle(hip, seed, B, A, nc):
    rng = np.random.default_rng(seed)
    pred = np.zeros((B, A, 5 + nc), np.float32)
    centers = rng.uniform(80, 560, (B, 9, 2)).astype(np.float32)
    idx = rng.integers(0, 9, (B, A))
    pred[..., 0:2] = np.take_along_axis(centers, idx[..., None].repeat(2, 2), 1) + rng.normal(0, 5, (B, A, 2))
    pred[..., 2:4] = 70 + rng.normal(0, 10, (B, A, 2))
    pred[..., 4] = rng.uniform(0, 1, (B, A)) ** 2
    pred[..., 5:] = rng.uniform(0, 1, (B, A, nc)) ** 2
    pred[:, ::11] = pred[:, 1::11][:, : pred[:, ::11].shape[1]] if A > 22 else pred[:, ::11]   # ties
    ref, rkeep = o_nms.non_max_suppression_ssod(pred, 0.1, 0.65)
    dets, counts, keep, ncand = _run(hip, pred, 0.1, 0.65)
    for i in range(B):
        assert counts[i] == ref[i].shape[0]
        assert np.array_equal(dets[i, :counts[i]], ref[i])
        assert np.array_equal(keep[i, :counts[i]], rkeep[i])


def test_nms_max_det_and_list_api(hip):
    from efficientteacher_amd.utils.general import non_max_suppression_ssod
    rng = np.random.default_rng(7)
    pred = np.zeros((2, 900, 9), np.float32)
    pred[..., 0:2] = rng.uniform(0, 640, (2, 900, 2))
    pred[..., 2:4] = rng.uniform(4, 30, (2, 900, 2))
    pred[..., 4] = rng.uniform(0.5, 1, (2, 900))
    pred[..., 5:] = rng.uniform(0.5, 1, (2, 900, 4))
    ref, _ = o_nms.non_max_suppression_ssod(pred, 0.1, 0.65, max_det=50)
    out = non_max_suppression_ssod(hip.t(pred), 0.1, 0.65, max_det=50)
    assert len(out) == 2
    for o, r in zip(out, ref):
        assert o.shape == (50, 8) and np.array_equal(o.cpu().numpy(), r)


NMS_SSOD_OPTIONS = {
    "classes": dict(classes=[1, 4]),
    "classes_agnostic": dict(classes=[0, 2, 5], agnostic=True),
    "multi_label": dict(multi_label=True),
    "multi_label_classes": dict(multi_label=True, classes=[3]),
    "labels": dict(labels=True),
    "labels_multi": dict(labels=True, multi_label=True, agnostic=True),
}


@pytest.mark.parametrize("variant", sorted(NMS_SSOD_OPTIONS))
def test_nms_ssod_optional_arguments_golden(hip, variant):
    """classes / multi_label / labels / agnostic of non_max_suppression_ssod (utils/general.py:887-992) vs the reference's own output
    (tests/golden/nms_ssod_options.npz, oracle/make_golden.py::case_nms_ssod_options): bit-exact rows, in order"""
    from efficientteacher_amd.utils.general import non_max_suppression_ssod
    g = golden("nms_ssod_options")
    kw = dict(NMS_SSOD_OPTIONS[variant])
    if kw.pop("labels", False):
        rows, cnt = g["apriori_rows"], g["apriori_counts"]
        offs = np.concatenate(([0], np.cumsum(cnt)))
        kw["labels"] = [hip.t(rows[offs[i]:offs[i + 1]]) for i in range(len(cnt))]
    out = non_max_suppression_ssod(hip.t(g["pred"]), float(g["thr"][0]), float(g["thr"][1]), **kw)
    w = 6 if kw.get("multi_label") else 8
    assert [o.shape[0] for o in out] == list(g[f"{variant}_counts"])
    got = np.concatenate([o.cpu().numpy().reshape(-1, w) for o in out], 0)
    assert np.array_equal(got, g[f"{variant}_dets"].reshape(-1, w))


def test_nms_val_path_apriori_labels(hip):
    """non_max_suppression(labels=...) (utils/general.py:1027-1034) vs the oracle restatement with the label rows appended"""
    from efficientteacher_amd.utils.general import non_max_suppression
    g = golden("nms_ssod_options")
    rows, cnt = g["apriori_rows"], g["apriori_counts"]
    offs = np.concatenate(([0], np.cumsum(cnt)))
    labels = [rows[offs[i]:offs[i + 1]] for i in range(len(cnt))]
    pred = g["pred"]
    nc = pred.shape[2] - 5
    ext = np.zeros((pred.shape[0], max(cnt), pred.shape[2]), np.float32)
    for i, l in enumerate(labels):
        ext[i, :len(l), :4] = l[:, 1:5]; ext[i, :len(l), 4] = 1.0
        ext[i, np.arange(len(l)), l[:, 0].astype(np.int64) + 5] = 1.0
    ref = o_nms.non_max_suppression(np.concatenate((pred, ext), 1), 0.1, 0.6, multi_label=True)
    out = non_max_suppression(hip.t(pred), 0.1, 0.6, multi_label=True, labels=[hip.t(l) for l in labels])
    for o, r in zip(out, ref):
        assert np.array_equal(o.cpu().numpy().reshape(-1, 6), r.reshape(-1, 6))


# ---- general non_max_suppression: the val.py path and boundary entry (8b) -----------------
def _run_general(hip, pred, ct, it, **kw):
    from efficientteacher_amd.utils.general import nms_padded
    dets, counts, keep, ncand = nms_padded(hip.t(pred), ct, it, **kw)
    return dets.cpu().numpy(), counts.cpu().numpy(), keep.cpu().numpy(), ncand.cpu().numpy()


@pytest.mark.parametrize("case", ["a", "b", "c", "empty", "tie06"])
def test_nms_val_golden(hip, case):
    """multi_label NMS vs the reference's own non_max_suppression(multi_label=True) output."""
    g = golden("nms")
    pred, (ct, it) = g[f"{case}_pred"], g[f"{case}_thr"]
    dets, counts, _, _ = _run_general(hip, pred, float(ct), float(it), multi_label=True)
    assert np.array_equal(counts, g[f"{case}_val_counts"])
    got = np.concatenate([dets[i, :c] for i, c in enumerate(counts)], 0).reshape(-1, 6)
    assert np.array_equal(got, g[f"{case}_val_dets"].reshape(-1, 6))    # bit exact rows


def _clustered(seed, B, A, nc, ties=True):
    rng = np.random.default_rng(seed)
    pred = np.zeros((B, A, 5 + nc), np.float32)
    centers = rng.uniform(80, 560, (B, 9, 2)).astype(np.float32)
    idx = rng.integers(0, 9, (B, A))
    pred[..., 0:2] = np.take_along_axis(centers, idx[..., None].repeat(2, 2), 1) + rng.normal(0, 5, (B, A, 2))
    pred[..., 2:4] = 70 + rng.normal(0, 10, (B, A, 2))
    pred[..., 4] = rng.uniform(0, 1, (B, A)) ** 2
    pred[..., 5:] = rng.uniform(0, 1, (B, A, nc)) ** 2
    if ties and A > 22:
        pred[:, ::11] = pred[:, 1::11][:, : pred[:, ::11].shape[1]]
    return pred


@pytest.mark.parametrize("multi", [False, True])
@pytest.mark.parametrize("seed,B,A,nc", [(0, 2, 700, 80), (1, 1, 64, 3), (2, 3, 333, 1), (3, 2, 1300, 20)])
def test_nms_general_vs_oracle(hip, seed, B, A, nc, multi):
    pred = _clustered(seed, B, A, nc)
    ref = o_nms.non_max_suppression(pred, 0.05, 0.6, multi_label=multi)
    dets, counts, _, _ = _run_general(hip, pred, 0.05, 0.6, multi_label=multi)
    for i in range(B):
        assert counts[i] == ref[i].shape[0]
        assert np.array_equal(dets[i, :counts[i]], ref[i])


def test_nms_general_classes_agnostic_and_list_api(hip):
    from efficientteacher_amd.utils.general import non_max_suppression
    pred = _clustered(5, 2, 500, 12)
    ref = o_nms.non_max_suppression(pred, 0.05, 0.5, multi_label=True, classes=[1, 4, 11], agnostic=True, max_det=40)
    out = non_max_suppression(hip.t(pred), 0.05, 0.5, classes=[1, 4, 11], agnostic=True, multi_label=True, max_det=40)
    assert len(out) == 2
    for o, r in zip(out, ref):
        assert o.shape[1] == 6 and np.array_equal(o.cpu().numpy(), r)


@pytest.mark.parametrize("max_nms", [97, 500, 2000])
def test_nms_general_max_nms_cut_with_ties(hip, max_nms):
    """More candidates than max_nms: the cut keeps the best scores, ties in candidate order (exact, via the
    two-level histogram select), including ties that straddle the cut."""
    pred = _clustered(9, 2, 600, 10)
    pred[:, 100:400, 4] = 0.5                   # many equal scores: obj and cls identical across anchors
    pred[:, 100:400, 5:] = pred[:, 100:101, 5:]
    ref = o_nms.non_max_suppression(pred, 0.01, 0.6, multi_label=True, max_nms=max_nms)
    dets, counts, _, ncand = _run_general(hip, pred, 0.01, 0.6, multi_label=True, max_nms=max_nms)
    assert (ncand == max_nms).all()
    for i in range(2):
        assert counts[i] == ref[i].shape[0]
        assert np.array_equal(dets[i, :counts[i]], ref[i])
