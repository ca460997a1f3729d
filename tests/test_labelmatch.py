"""LabelMatch (reference utils/labelmatch.py): device pseudo labels + score log + per-class thresholds at the
end of an epoch, vs the reference-run golden (tests/golden/labelmatch.npz, `python -m oracle.make_golden labelmatch`)."""
import numpy as np
import pytest
import torch

from oracle.make_golden import labelmatch_pred
from tests.conftest import golden


def _cfg(g):
    import os
    from efficientteacher_amd.configs import get_cfg
    from tests.conftest import ROOT
    nc = int(g["nc"])
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "efficientteacher_amd/configs/ssod/coco-standard/yolov5l_coco_ssod_10_percent.yaml"))
    cfg.merge_from_list(["Dataset.nc", nc, "SSOD.pseudo_label_type", "LabelMatch", "SSOD.resample_low_percent", float(g["thr"][4]),
                         "SSOD.resample_high_percent", float(g["thr"][5]), "Dataset.names", [str(i) for i in range(nc)]])
    return cfg


def test_two_epochs_against_the_reference(hip):
    from efficientteacher_amd.utils.labelmatch import LabelMatch
    g = golden("labelmatch")
    nc = int(g["nc"])
    cfg = _cfg(g)
    assert np.allclose([cfg.SSOD.nms_conf_thres, cfg.SSOD.nms_iou_thres, cfg.SSOD.ignore_thres_low, cfg.SSOD.ignore_thres_high], g["thr"][:4])
    B, A = (int(v) for v in g["BA"])
    H, W = (int(v) for v in g["hw"])
    rng = np.random.default_rng(int(g["seed"]))
    lm = LabelMatch(cfg, 100, 5, cls_ratio_gt=np.full(nc, 1.0 / nc))
    M_s = hip.t(g["M_s"])
    imgs = torch.zeros(B, 3, H, W)
    step = 0
    for epoch, nb in enumerate((3, 2)):
        for _ in range(nb):
            pred = labelmatch_pred(rng, B, A, nc)
            t, invalid = lm.create_pseudo_label_online_with_gt(hip.t(pred), imgs, M_s, imgs)
            ref = g[f"targets{step}"]
            assert invalid == (ref.shape[0] == 0)
            got = t.cpu().numpy()
            assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-12, step
            lm.update(hip.t(np.array([[0, 1, .5, .5, .1, .1], [1, 3, .5, .5, .1, .1]], np.float32)), 2, B)
            step += 1
        lm.update_epoch_cls_thr(epoch)
        assert np.array_equal(np.array(lm.cls_thr_high), g[f"thr_high{epoch}"]), (epoch, lm.cls_thr_high)
        assert np.array_equal(np.array(lm.cls_thr_low), g[f"thr_low{epoch}"]), (epoch, lm.cls_thr_low)
        assert lm.count == 0 and int(lm._log[2].item()) == 0


def test_score_log_overflow_is_an_error(hip):
    from efficientteacher_amd.utils.labelmatch import LabelMatch
    g = golden("labelmatch")
    nc = int(g["nc"])
    lm = LabelMatch(_cfg(g), 1, 5, cls_ratio_gt=np.full(nc, 1.0 / nc), score_log_capacity=16)
    B, A = (int(v) for v in g["BA"])
    pred = labelmatch_pred(np.random.default_rng(1), B, A, nc)
    lm.create_pseudo_label_padded(hip.t(pred), hip.t(g["M_s"]), 640, 640)
    with pytest.raises(RuntimeError, match="overflow"):
        lm.update_epoch_cls_thr(0)


def test_thresholds_reach_the_unsupervised_loss(hip):
    """after_epoch (ssod_trainer.py:319-323): the per-class lists replace the loss's thresholds and change which pseudo labels
    are reliable / uncertain / dropped"""
    from efficientteacher_amd import ops
    t9 = np.zeros((4, 9), np.float64)
    t9[:, 1] = [0, 0, 1, 1]
    t9[:, 2:6] = 0.5
    t9[:, 6] = [0.5, 0.2, 0.5, 0.2]
    t9[:, 7:9] = 0.5
    valid = hip.t(np.ones(4, np.uint8))
    table = ops.select_targets(hip.t(t9), valid, [0.1, 0.3], [0.4, 0.6], 2, True).cpu().numpy()
    assert list(table[:, 7].astype(int) & 3) == [1, 2, 2, 0]


def test_trainer_step_and_epoch_end(hip):
    """SSODTrainer with SSOD.pseudo_label_type = LabelMatch: steps log scores on the device, after_epoch moves the loss's
    thresholds (ssod_trainer.py:70-71, :319-323, :616-617)"""
    import os
    from efficientteacher_amd.configs import get_cfg
    from efficientteacher_amd.trainer import SSODTrainer
    from tests.conftest import ROOT
    from tests.test_ssod_step import YAML
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, YAML))
    cfg.merge_from_list(["Model.width_multiple", 0.125, "Model.depth_multiple", 0.33, "SSOD.pseudo_label_type", "LabelMatch",
                         "SSOD.resample_low_percent", 0.5])
    cfg.freeze()
    g = golden("ssod_step")
    nc = cfg.Dataset.nc
    t = SSODTrainer(cfg, hip.device, nb=1000, target_data_len=8, label_num_per_image=3, cls_ratio_gt=np.full(nc, 1.0 / nc))
    t.model.set_compute_dtype(torch.float32)
    with torch.no_grad():          # let the random teacher detect something
        for mi in t.model.head.m:
            b = mi.bias.view(t.model.head.na, -1)
            b[:, 4] += 6.0
            b[:, 5:] += 3.5
            b[:, 5 + 7] += 2.0
    t.build_optimizer(cfg)
    from efficientteacher_amd.utils.torch_utils import ModelEMA
    t.ema = ModelEMA(t.model)
    t.semi_ema = None
    lo0 = list(t.compute_un_sup_loss.ignore_thres_low)
    for ni in (500, 501):
        items = t.train_instance(hip.t(g["imgs"]), hip.t(g["targets"]), None, hip.t(g["u_str"]), hip.t(g["u_ori"]), None,
                                 hip.t(g["M_s"]), ni)
    assert all(np.isfinite(float(v)) for v in items.values())
    n_logged = int(t.pseudo_label_creator._log[2].item())
    assert n_logged > 0 and t.pseudo_label_creator.count == 2 * g["imgs"].shape[0]
    t.after_epoch(0)
    lm = t.pseudo_label_creator
    assert t.compute_un_sup_loss.ignore_thres_low is lm.cls_thr_low and t.compute_un_sup_loss.ignore_thres_high is lm.cls_thr_high
    assert list(lm.cls_thr_low) != lo0 and int(lm._log[2].item()) == 0
    assert lm.cls_num_total.sum() == n_logged
    items = t.train_instance(hip.t(g["imgs"]), hip.t(g["targets"]), None, hip.t(g["u_str"]), hip.t(g["u_ori"]), None,
                             hip.t(g["M_s"]), 502)
    assert all(np.isfinite(float(v)) for v in items.values())


def test_thresholds_live_in_one_persistent_device_tensor(hip):
    """the per-class thresholds of select_targets were cached BY VALUE; a captured step graph kept reading the
    tensor of the old values after LabelMatch rewrote the lists (and a cache eviction could free it).  They now live in ONE
    tensor per loss object that is refreshed in place: same address before and after a change, new values in effect."""
    from efficientteacher_amd import ops
    th = ops.DeviceThresholds()
    t9 = np.zeros((2, 9)); t9[:, 1] = [0, 1]; t9[:, 2:6] = 0.5; t9[:, 6] = [0.5, 0.5]; t9[:, 7] = 0.9; t9[:, 8] = 0.9
    valid = hip.t(np.ones(2, np.uint8))
    a = ops.select_targets(hip.t(t9), valid, [0.1, 0.1], [0.4, 0.6], 2, True, thresholds=th).cpu().numpy()
    p0 = th.refresh([0.1, 0.1], [0.4, 0.6], hip.device).data_ptr()
    b = ops.select_targets(hip.t(t9), valid, [0.1, 0.1], [0.6, 0.4], 2, True, thresholds=th).cpu().numpy()
    assert th.refresh([0.1, 0.1], [0.6, 0.4], hip.device).data_ptr() == p0
    # flags (column 7): class 0 was reliable (0.5 >= 0.4) and becomes uncertain (0.5 < 0.6); class 1 the other way round
    assert a[0, 7] != b[0, 7] and a[1, 7] != b[1, 7] and a[0, 7] == b[1, 7]
