"""Row (b) of SURVEY.md section 8: the reference's OWN epoch loop (Trainer.before_epoch / train_in_epoch,
trainer/trainer.py:358,406; SSODTrainer.train_in_epoch -> train_with_unlabeled, trainer/ssod_trainer.py:295,682) drives
this package's hot path through ``efficientteacher_amd.trainer.adapters.hot_path_trainers()``.

Needs the reference tree (build container only: it is imported live through oracle/ref_loader.py, nothing is copied);
on the GPU box, where /root/reference does not exist, the module is skipped.  Kernels run in the SIMT emulator.
The data loaders are replaced as SURVEY.md section 8(c) describes (lists of the reference's batch tuples)."""
import os
import tempfile

import numpy as np
import pytest
import torch

from oracle import ref_loader
from tests.conftest import ROOT

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="reference tree not present (build container only)")
SSOD_YAML = "configs/ssod/coco-standard/yolov5l_coco_ssod_10_percent.yaml"


class _Loader(list):
    num_workers = 0
    sampler = None


class _Dataset:
    mosaic = True

    def __init__(self, labels):
        self.labels = labels
        self.cls_ratio_gt = None
        self.label_num_per_image = None

    def __len__(self):
        return 4


def _batches(rng, n, B, S, unlabeled=False):
    out = _Loader()
    for _ in range(n):
        imgs = torch.from_numpy(rng.integers(0, 256, (B, 3, S, S), dtype=np.uint8))
        t = []
        for b in range(B):
            k = int(rng.integers(1, 4))
            t.append(np.concatenate((np.full((k, 1), b), rng.integers(0, 80, (k, 1)), rng.uniform(0.3, 0.7, (k, 2)),
                                     rng.uniform(0.1, 0.4, (k, 2))), 1))
        targets = torch.from_numpy(np.concatenate(t, 0).astype(np.float32))
        if not unlabeled:
            out.append((imgs, targets, [f"img{b}.jpg" for b in range(B)], None))
        else:
            M = torch.zeros(B, 13, dtype=torch.float64)
            for b in range(B):
                M[b] = torch.tensor([b, 1, 0, 0, 0, 1, 0, 0, 0, 1, 1.0, 0, b % 2], dtype=torch.float64)
            out.append((imgs, targets, [f"u{b}.jpg" for b in range(B)], None, imgs.clone(), M))
    return out


def _cfg(save_dir, ssod, extra=()):
    cfg = ref_loader.get_cfg(SSOD_YAML, ["Model.width_multiple", 0.125, "Model.depth_multiple", 0.33, "Dataset.batch_size", 2,
                                         "Dataset.img_size", 32, "save_dir", save_dir, "noval", True, "nosave", True,
                                         "epochs", 2, "SSOD.train_domain", bool(ssod), "device", "cpu", "Dataset.workers", 0,
                                         "hyp.burn_epochs", 0] + list(extra))
    cfg.freeze()
    return cfg


@pytest.fixture
def ref_callbacks():
    ref_loader.load()
    from utils.callbacks import Callbacks
    cb = Callbacks()
    cb._callbacks = {k: [] for k in cb._callbacks}     # the hook table is a CLASS attribute there: drop other tests' loggers
    return cb


def _mk(base, rng, ssod):
    class T(base):
        def build_dataloader(self, cfg, callbacks):          # no image files on disk
            self.imgsz = cfg.Dataset.img_size
            self.train_loader = _batches(rng, 2, 2, cfg.Dataset.img_size)
            self.dataset = _Dataset([np.array([[3, .5, .5, .2, .2]], np.float32)] * 4)
            self.nb = len(self.train_loader)
            self.no_aug_epochs = cfg.hyp.no_aug_epochs
            if ssod:
                self.unlabeled_dataloader = _batches(rng, 2, 2, cfg.Dataset.img_size, unlabeled=True)
                self.unlabeled_dataset = _Dataset([])
                self.cls_ratio_gt = np.full(cfg.Dataset.nc, 1.0 / cfg.Dataset.nc)
                self.label_num_per_image = 2
    return T


def test_reference_epoch_loop_drives_the_hot_path_supervised(emu, ref_callbacks):
    from efficientteacher_amd.trainer.adapters import hot_path_trainers
    from efficientteacher_amd.models.detector.yolo import Model as EtModel
    from efficientteacher_amd.optim import FlatSGD
    Trainer, _ = hot_path_trainers()
    rng = np.random.default_rng(0)
    with tempfile.TemporaryDirectory() as d:
        t = _mk(Trainer, rng, False)(_cfg(d, False), torch.device("cpu"), ref_callbacks, -1, -1, 1)
        assert isinstance(t.model, EtModel) and isinstance(t.optimizer, FlatSGD)
        import trainer.trainer as ref_mod
        assert isinstance(t, ref_mod.Trainer)                      # the reference's class: its loop, its loggers, its checkpoints
        p0 = t.model.flat_state().params.clone()
        t.last_opt_step = -1
        t.plots = False                                            # the reference's plotting thread needs an older PIL
        t.before_epoch()                                           # reference code: meters, warm-up length, first logging forward
        assert t.nw == -1 or t.nw >= 0
        t.train_in_epoch(ref_callbacks)                            # reference code: two iterations + scheduler.step()
        assert torch.isfinite(t.model.flat_state().params).all()
        assert not torch.equal(p0, t.model.flat_state().params)
        assert len(t.meter.meters) >= 3 and all(np.isfinite(v) for v in t.meter.get_avg())
        assert t.ema.updates >= 1


def test_reference_epoch_loop_in_fp16_mode_uses_the_device_grad_scaler(emu, ref_callbacks):
    """hot_path_trainers(compute_dtype=torch.float16): the reference's own update_optimizer (scaler.scale(loss).backward();
    scaler.step(optimizer); scaler.update() -- trainer.py:399-401) drives the device-resident scaler: every iteration either moves
    the parameters or halves the scale, and nothing becomes non-finite"""
    from efficientteacher_amd.trainer.adapters import AmpScaler, hot_path_trainers
    Trainer, _ = hot_path_trainers(compute_dtype=torch.float16)
    rng = np.random.default_rng(0)
    with tempfile.TemporaryDirectory() as d:
        t = _mk(Trainer, rng, False)(_cfg(d, False), torch.device("cpu"), ref_callbacks, -1, -1, 1)
        assert t.model.flat_state().compute_dtype == torch.float16 and isinstance(t.scaler, AmpScaler)
        s0 = t.scaler.get_scale()
        p0 = t.model.flat_state().params.clone()
        t.last_opt_step = -1
        t.plots = False
        t.before_epoch()
        t.train_in_epoch(ref_callbacks)
        assert torch.isfinite(t.model.flat_state().params).all()
        moved = not torch.equal(p0, t.model.flat_state().params)
        assert moved or t.scaler.get_scale() < s0, (moved, s0, t.scaler.get_scale())
        assert t.scaler.get_scale() in (s0, s0 / 2, s0 / 4)
        sd = t.scaler.state_dict()
        t.scaler.load_state_dict(sd)
        assert t.scaler.get_scale() == float(sd["state"][0])


def test_reference_epoch_loop_drives_the_hot_path_ssod(emu, ref_callbacks):
    from efficientteacher_amd.trainer.adapters import hot_path_trainers
    from efficientteacher_amd.models.detector.yolo_ssod import Model as EtModel
    _, SSODTrainer = hot_path_trainers()
    rng = np.random.default_rng(1)
    with tempfile.TemporaryDirectory() as d:
        t = _mk(SSODTrainer, rng, True)(_cfg(d, True), torch.device("cpu"), ref_callbacks, -1, -1, 1)
        assert isinstance(t.model, EtModel)
        import trainer.ssod_trainer as ref_mod
        assert isinstance(t, ref_mod.SSODTrainer)
        with torch.no_grad():            # a random-init teacher detects nothing: bump the objectness / class biases
            for mi in t.model.head.m:
                b = mi.bias.view(t.model.head.na, -1)
                b[:, 4] += 6.0
                b[:, 5:] += 3.5
        t.model.flat_state().mark_weights_changed()
        from efficientteacher_amd.utils.torch_utils import ModelEMA
        t.ema = ModelEMA(t.model)
        p0 = t.model.flat_state().params.clone()
        e0 = t.ema.ema.flat_state().params.clone()
        t.last_opt_step = -1
        t.plots = False                                            # the reference's plotting thread needs an older PIL
        t.target_with_gt = False                                   # the recipes' setting; the LabelMatch test below keeps with_gt
        t.before_epoch()
        t.train_in_epoch(ref_callbacks)                            # -> the reference's train_with_unlabeled -> OUR train_instance
        assert torch.isfinite(t.model.flat_state().params).all()
        assert not torch.equal(p0, t.model.flat_state().params)
        assert not torch.equal(e0, t.ema.ema.flat_state().params)   # the EMA teacher followed
        names = set(t.meter.meters.keys())
        assert {"box", "obj", "cls", "ss_box", "ss_obj", "ss_cls"} <= names, names
        assert float(t.meter.meters["ss_obj"].avg) > 0             # pseudo labels reached the unsupervised loss
        # the progress-bar statistics of ssod_trainer.py:657-673, against the reference's own routine on the last step's labels
        assert {"tp", "fp_cls", "fp_loc", "pse_num", "gt_num"} <= names, names
        from utils.self_supervised_utils import check_pseudo_label
        t9, valid = t._last_pseudo
        rows = t9[valid.bool()].float().cpu()
        want = check_pseudo_label(rows, ignore_thres_low=t.compute_un_sup_loss.ignore_thres_low,
                                  ignore_thres_high=t.compute_un_sup_loss.ignore_thres_high, batch_size=t.batch_size)
        got = [t.meter.meters[k].val for k in ("tp", "fp_loc", "pse_num", "gt_num")]
        got, want = [float(np.asarray(x).reshape(-1)[0]) for x in got], [float(x) for x in want]
        assert rows.shape[0] > 0 and np.allclose(got, want, rtol=1e-12, atol=0), (got, want)


def test_reference_epoch_loop_with_labelmatch(emu, ref_callbacks):
    """SSOD.pseudo_label_type = LabelMatch: the reference constructs its own LabelMatch (ssod_trainer.py:70-71), the adapter
    swaps in the device one; the reference's train_instance bookkeeping (:616-617) and after_epoch threshold hand-over
    (:319-323) then run against it"""
    from efficientteacher_amd.trainer.adapters import hot_path_trainers
    from efficientteacher_amd.utils.labelmatch import LabelMatch
    _, SSODTrainer = hot_path_trainers()
    rng = np.random.default_rng(2)
    with tempfile.TemporaryDirectory() as d:
        cfg = _cfg(d, True, ["SSOD.pseudo_label_type", "LabelMatch", "SSOD.resample_low_percent", 0.5])
        t = _mk(SSODTrainer, rng, True)(cfg, torch.device("cpu"), ref_callbacks, -1, -1, 1)
        assert isinstance(t.pseudo_label_creator, LabelMatch)
        with torch.no_grad():
            for mi in t.model.head.m:
                b = mi.bias.view(t.model.head.na, -1)
                b[:, 4] += 6.0
                b[:, 5:] += 3.5
        t.model.flat_state().mark_weights_changed()
        from efficientteacher_amd.utils.torch_utils import ModelEMA
        t.ema = ModelEMA(t.model)
        t.last_opt_step = -1
        t.plots = False
        t.before_epoch()
        t.train_in_epoch(ref_callbacks)
        lm = t.pseudo_label_creator
        assert int(lm._log[2].item()) > 0 and lm.count > 0
        lm.update_epoch_cls_thr(0)                                  # what after_epoch (:320) calls
        t.compute_un_sup_loss.ignore_thres_high = lm.cls_thr_high   # :321-322
        t.compute_un_sup_loss.ignore_thres_low = lm.cls_thr_low
        assert len(lm.cls_thr_low) == cfg.Dataset.nc and max(lm.cls_thr_low) > cfg.SSOD.ignore_thres_low


def test_reference_epoch_loop_drives_the_v8_path(emu, ref_callbacks):
    """configs/sup/public/yolov8m_coco.yaml (C2f / DFL head / TAL loss) under the reference's own Trainer loop: the reference
    cannot build its ComputeTalLoss (its gfocal_loss module is missing), the adapter's build_ddp_model supplies this package's"""
    from efficientteacher_amd.trainer.adapters import hot_path_trainers
    from efficientteacher_amd.models.loss import ComputeTalLoss
    Trainer, _ = hot_path_trainers()
    rng = np.random.default_rng(3)
    with tempfile.TemporaryDirectory() as d:
        cfg = ref_loader.get_cfg("configs/sup/public/yolov8m_coco.yaml",
                                 ["Model.width_multiple", 0.125, "Model.depth_multiple", 0.33, "Dataset.batch_size", 2,
                                  "Dataset.img_size", 32, "save_dir", d, "noval", True, "nosave", True, "epochs", 2, "device", "cpu",
                                  "Dataset.workers", 0])
        cfg.freeze()
        t = _mk(Trainer, rng, False)(cfg, torch.device("cpu"), ref_callbacks, -1, -1, 1)
        assert isinstance(t.compute_loss, ComputeTalLoss)
        p0 = t.model.flat_state().params.clone()
        t.last_opt_step = -1
        t.plots = False
        t.before_epoch()
        t.train_in_epoch(ref_callbacks)
        assert torch.isfinite(t.model.flat_state().params).all() and not torch.equal(p0, t.model.flat_state().params)
        assert all(np.isfinite(v) for v in t.meter.get_avg())


def test_burn_in_epoch_then_ssod_epoch(emu, ref_callbacks):
    """hyp.burn_epochs = 1 (the 1 / 2 / 5 % COCO recipes burn in for 220 epochs): epoch 0 runs the reference's supervised
    train_without_unlabeled over the hot path, epoch 1 creates the semi-supervised EMA -- THIS package's arena-aware class, not
    the reference's per-tensor one -- and runs the SSOD step"""
    from efficientteacher_amd.trainer.adapters import hot_path_trainers
    from efficientteacher_amd.utils.torch_utils import CosineEMA as EtCosineEMA, ModelEMA
    _, SSODTrainer = hot_path_trainers()
    rng = np.random.default_rng(4)
    with tempfile.TemporaryDirectory() as d:
        t = _mk(SSODTrainer, rng, True)(_cfg(d, True, ["hyp.burn_epochs", 1]), torch.device("cpu"), ref_callbacks, -1, -1, 1)
        assert t.semi_ema is None
        with torch.no_grad():
            for mi in t.model.head.m:
                b = mi.bias.view(t.model.head.na, -1)
                b[:, 4] += 6.0
                b[:, 5:] += 3.5
        t.model.flat_state().mark_weights_changed()
        t.ema = ModelEMA(t.model)
        t.last_opt_step = -1
        t.plots = False
        t.epoch = 0
        t.before_epoch()
        p0 = t.model.flat_state().params.clone()
        t.train_in_epoch(ref_callbacks)                   # burn-in: supervised only
        assert t.semi_ema is None and not torch.equal(p0, t.model.flat_state().params)
        assert "ss_obj" not in t.meter.meters or float(t.meter.meters["ss_obj"].avg) == 0
        t.epoch = 1
        t.before_epoch()
        t.train_in_epoch(ref_callbacks)                   # first SSOD epoch
        assert isinstance(t.semi_ema, EtCosineEMA)
        assert float(t.meter.meters["ss_obj"].avg) > 0
        assert torch.isfinite(t.model.flat_state().params).all()


def test_reference_after_epoch_saves_a_reference_checkpoint_and_resumes(emu, ref_callbacks):
    """the reference's own after_epoch (ssod_trainer.py:319-411) pickles `deepcopy(model).half()`: with the hot-path model in
    place the file must still be a REFERENCE checkpoint (its Model class, fp16 tensors, optimizer state), and a new trainer
    built with cfg.weights = last.pt starts from those weights"""
    from efficientteacher_amd.trainer.adapters import hot_path_trainers
    _, SSODTrainer = hot_path_trainers()
    rng = np.random.default_rng(6)
    with tempfile.TemporaryDirectory() as d:
        t = _mk(SSODTrainer, rng, True)(_cfg(d, True, ["nosave", False]), torch.device("cpu"), ref_callbacks, -1, -1, 1)
        t.last_opt_step = -1
        t.plots = False
        t.before_epoch()
        t.train_in_epoch(ref_callbacks)
        t.results, t.best_fitness, t.lr = (0, 0, 0, 0, 0, 0, 0), 0.0, [0.0, 0.0, 0.0]
        t.after_epoch(ref_callbacks, None)                       # noval: bookkeeping + checkpoint files only
        ck = torch.load(str(t.last), map_location="cpu", weights_only=False)
        assert type(ck["model"]).__module__ == "models.detector.yolo_ssod" and type(ck["ema"]).__module__ == "models.detector.yolo_ssod"
        assert next(ck["model"].parameters()).dtype == torch.float16
        assert "flat_momentum" in ck["optimizer"] and ck["epoch"] == 0
        msd = {k: v.detach().cpu() for k, v in t.model.state_dict().items()}
        for k, v in ck["model"].float().state_dict().items():
            if v.is_floating_point():
                assert (v - msd[k].float()).abs().max() <= 1e-3 * max(1.0, msd[k].abs().max().item()), k
        # a fresh trainer picks the file up through cfg.weights (trainer.py:127-144 semantics)
        t2 = _mk(SSODTrainer, rng, True)(_cfg(d, True, ["weights", str(t.last)]), torch.device("cpu"), ref_callbacks, -1, -1, 1)
        k = "backbone.stage1.conv.weight"
        assert (t2.model.state_dict()[k].cpu() - ck["model"].state_dict()[k]).abs().max() == 0


def test_reference_val_run_over_the_hot_path_model(emu, ref_callbacks):
    """the reference's own val.run (val.py:149-400: its loop, its NMS, its metrics) fed with the hot-path model, against the
    same call on the reference's Model carrying the same weights (obtained by pickling: utils/checkpoint.reduce_model): the
    metric tuples agree.  (The loss entries are 0 on both sides: `if outputs is not list` at val.py:305 is always true, so the
    reference never reaches its compute_loss call.)"""
    import io
    from copy import deepcopy
    from pathlib import Path
    import val as ref_val
    from models.loss.loss import ComputeLoss as RefComputeLoss
    from efficientteacher_amd.trainer.adapters import hot_path_trainers
    _, SSODTrainer = hot_path_trainers()
    rng = np.random.default_rng(8)
    with tempfile.TemporaryDirectory() as d:
        cfg = _cfg(d, True)
        t = _mk(SSODTrainer, rng, True)(cfg, torch.device("cpu"), ref_callbacks, -1, -1, 1)
        with torch.no_grad():
            for mi in t.model.head.m:
                b = mi.bias.view(t.model.head.na, -1)
                b[:, 4] += 4.0
                b[:, 5:] += 2.0
        t.model.flat_state().mark_weights_changed()
        buf = io.BytesIO()
        torch.save(t.model, buf)
        buf.seek(0)
        ref_model = torch.load(buf, map_location="cpu", weights_only=False).float()
        assert type(ref_model).__module__ == "models.detector.yolo_ssod"
        # labels = a few of the reference model's own confident detections, so that precision / recall / mAP are not all zero
        from utils.general import non_max_suppression as ref_nms, xyxy2xywh
        loader = []
        ref_model.eval()
        for bi in range(2):
            imgs = torch.from_numpy(rng.integers(0, 256, (2, 3, 64, 64), dtype=np.uint8))
            with torch.no_grad():
                z = ref_model(imgs.float() / 255.0)[0][0]
            rows = []
            for i, det in enumerate(ref_nms(z, 0.005, 0.45, max_det=3)):
                for *xyxy, conf, c in det.tolist():
                    rows.append([i, c, *(xyxy2xywh(torch.tensor([xyxy])) / 64.0)[0].tolist()])
            tg = torch.tensor(rows, dtype=torch.float32).reshape(-1, 6)
            loader.append((imgs, tg, [f"a{bi}.jpg", f"b{bi}.jpg"], [((64, 64), ((1.0, 1.0), (0.0, 0.0)))] * 2))
        assert sum(x[1].shape[0] for x in loader) >= 4
        data = {'nc': 80, 'names': cfg.Dataset.names, 'val': 'x'}

        def run(model, closs):
            return ref_val.run(data, batch_size=2, imgsz=64, model=model, conf_thres=0.001, single_cls=False,
                               dataloader=[(a.clone(), b.clone(), c, s) for a, b, c, s in loader], save_dir=Path(d), plots=False,
                               callbacks=ref_callbacks, compute_loss=closs, num_points=0, val_ssod=True, val_kp=False)[0]
        mine = run(deepcopy(t.model), t.compute_loss)
        ref = run(ref_model, RefComputeLoss(ref_model, cfg))
        assert ref[2] > 0.5, ref                                   # mAP@.5 of a model scored against its own detections
        assert np.allclose(mine[:4], ref[:4], atol=2e-3), (mine, ref)
        assert np.allclose(mine[4:], ref[4:], rtol=2e-4, atol=1e-6), (mine, ref)
