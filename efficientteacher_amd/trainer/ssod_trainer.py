"""Efficient-Teacher SSOD trainer core (host-side mirror of reference trainer/ssod_trainer.py:
build_model :96, update_optimizer :458-488, split_predict_and_feature :568, train_instance :587-680).

One ``train_instance`` = EMA-teacher inference on the weak view -> NMS + pseudo-label transform ->
student forward on cat(labelled, unlabelled strong view) -> ComputeLoss + ComputeStudentMatchLoss ->
backward -> (every ``accumulate`` iterations) SGD step, ModelEMA update, semi-EMA update.
Everything between the two image batches arriving and the optimizer step is device-resident and
stream-ordered: no ``.cpu()``, no per-detection python loops, no deepcopy of the image batches.
"""
from contextlib import nullcontext as _nullcontext

import torch

from ..models.loss import DomainLoss, TargetLoss, build_ssod_loss
from ..parallel import FlatDataParallel
from ..utils.self_supervised_utils import FairPseudoLabel
from ..utils.torch_utils import CosineEMA, SemiSupModelEMA
from .trainer import Trainer


class SSODTrainer(Trainer):
    MODEL_MODULE = "efficientteacher_amd.models.detector.yolo_ssod"

    def __init__(self, cfg, device, callbacks=None, LOCAL_RANK=-1, RANK=-1, WORLD_SIZE=1, nb=1000, target_data_len=None,
                 label_num_per_image=None, cls_ratio_gt=None, amp_dtype=None):
        """target_data_len / label_num_per_image / cls_ratio_gt: what the reference reads off its datasets for LabelMatch
        (ssod_trainer.py:71, :226-227); this core has no data loaders, the caller passes them."""
        self.cfg = cfg
        self._amp_dtype_arg = amp_dtype            # Trainer.set_env: bf16 (default) | fp16 (the reference's autocast dtype) | fp32
        self.set_env(cfg, device, LOCAL_RANK, RANK, WORLD_SIZE, callbacks, nb)
        self.build_model(cfg, device)
        self.build_optimizer(cfg)
        if cfg.SSOD.pseudo_label_type == 'FairPseudoLabel':
            self.pseudo_label_creator = FairPseudoLabel(cfg)
        elif cfg.SSOD.pseudo_label_type == 'LabelMatch':
            from ..utils.labelmatch import LabelMatch
            if cls_ratio_gt is None:
                raise ValueError("LabelMatch needs target_data_len, label_num_per_image and cls_ratio_gt (ssod_trainer.py:71)")
            self.pseudo_label_creator = LabelMatch(cfg, int(target_data_len / WORLD_SIZE), label_num_per_image,
                                                   cls_ratio_gt=cls_ratio_gt)
        else:
            raise NotImplementedError(f"SSOD.pseudo_label_type {cfg.SSOD.pseudo_label_type}")
        self.build_ddp_model(cfg, device)

    def after_epoch(self, epoch=None):
        """the LabelMatch part of ssod_trainer.py:319-323: re-estimate the per-class thresholds, hand them to the loss"""
        epoch = self.epoch if epoch is None else epoch
        if self.cfg.SSOD.pseudo_label_type == 'LabelMatch' and epoch >= self.cfg.SSOD.dynamic_thres_epoch:
            self.pseudo_label_creator.update_epoch_cls_thr(epoch - self.start_epoch)
            self.compute_un_sup_loss.ignore_thres_high = self.pseudo_label_creator.cls_thr_high
            self.compute_un_sup_loss.ignore_thres_low = self.pseudo_label_creator.cls_thr_low

    def set_env(self, cfg, device, LOCAL_RANK, RANK, WORLD_SIZE, callbacks, nb=1000):
        super().set_env(cfg, device, LOCAL_RANK, RANK, WORLD_SIZE, callbacks, nb)
        self.epoch_adaptor = cfg.SSOD.epoch_adaptor
        self.da_loss_weights = cfg.SSOD.da_loss_weights
        self.cosine_ema = cfg.SSOD.cosine_ema
        self.fixed_accumulate = cfg.SSOD.fixed_accumulate
        self.extra_teacher_models = []
        self.teacher_pred_hook = None      # optional callable(teacher_pred) -> teacher_pred (bench: synthetic scores)
        self.overlap_teacher = True        # teacher forward + pseudo labels on a second stream
        self.teacher_after = "p3"          # "" (start of the step) | "p1" | "p2" | "p3" | "p4": see _train_instance_eager
        self._side = None
        # the step as one captured HIP graph (trainer/graph_step.py), opt-in: ET_STEP_GRAPH=1 or use_graph=True.  Measured on
        # MI355X (profiles/r02_graph_step_timing.txt): issuing the ~750 launches of an eager step takes the host 18-24 ms, a
        # replay 3 ms, and both finish in 64-65 ms -- at 32+32 images the step is GPU-bound, so eager stays the default and the
        # graph is for small per-GPU batches.  The first `graph_warmup` steps always run eagerly (first-step flags, allocator,
        # lazily created streams).
        import os
        self.use_graph = os.environ.get("ET_STEP_GRAPH", "0") == "1"
        self.graph_warmup = 3
        self._eager_steps = 0
        self._capturing = False
        self._graph = None
        self.graph_error = None
        self._strong_view = None           # utils/augment.StrongViewGenerator, created on first use (weak-view-only batches)

    def _side_stream(self):
        if self._side is None:
            # (a CU-masked stream, hipExtStreamCreateWithCUMask with 64 / 96 / 128 / 192 of the 256 CUs, was measured in r05: the STEP got
            # 9-22 ms slower with every mask, main-stream kernels included -- profiles/r05_teacher_cu_mask_ab.txt -- and was removed)
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def build_model(self, cfg, device):
        super().build_model(cfg, device)            # student + ModelEMA (the teacher is self.ema.ema)
        if cfg.hyp.burn_epochs > 0:
            self.semi_ema = None
        elif self.cosine_ema:
            self.semi_ema = CosineEMA(self.ema.ema, decay_start=cfg.SSOD.ema_rate, total_epoch=self.epochs)
        else:
            self.semi_ema = SemiSupModelEMA(self.ema.ema, cfg.SSOD.ema_rate)

    def build_ddp_model(self, cfg, device):
        super().build_ddp_model(cfg, device)
        self.compute_un_sup_loss = build_ssod_loss(self.model, cfg)
        self.domain_loss = DomainLoss()
        self.target_loss = TargetLoss()

    # main stream joins the teacher stream in front of the UNSUPERVISED loss (True) or right behind the student's forward (False: until r06).  Class attribute: an A/B switch
    join_teacher_late = True

    def update_optimizer(self, loss, ni):
        self.scaler.scale(loss).backward()     # ssod_trainer.py:469 (identity unless the compute dtype is fp16)
        if self._capturing:                # graph capture: launches only; the host-side schedule runs before each replay
            if isinstance(self.model, FlatDataParallel):
                self.model.reduce_gradients()      # the remaining chunks + the waits of the async all-reduces, captured too
            self.scaler.step(self.optimizer)
            self.scaler.update()
            self.optimizer.zero_grad()
            self.ema.update(self.model)
            if self.semi_ema:
                self.semi_ema.update(self.ema.ema)
            return
        if isinstance(self.model, FlatDataParallel):
            self.model.reduce_gradients()
        self.accumulate = 1 if self.fixed_accumulate else max(round(64 / self.batch_size), 1)
        self._warmup(ni, 1 if self.fixed_accumulate else 64 / self.batch_size)
        if ni - self.last_opt_step >= self.accumulate:
            self.scaler.step(self.optimizer)       # ssod_trainer.py:482-483
            self.scaler.update()
            self.optimizer.zero_grad()
            self.ema.update(self.model)
            if self.semi_ema:
                self.semi_ema.update(self.ema.ema)
            self.last_opt_step = ni

    @staticmethod
    def split_predict_and_feature(total_pred, total_feature, n_img):
        sup_feature = [f[:n_img] for f in total_feature]
        un_sup_feature = [f[n_img:] for f in total_feature]
        if isinstance(total_pred, tuple) and len(total_pred) == 3 and isinstance(total_pred[0], (list, tuple)):
            # anchor-free head (YoloV8Detect train output): (feats, cls (B,A,nc), reg (B,A,4*(reg_max+1)))
            feats, cls, reg = total_pred
            return (([f[:n_img] for f in feats], cls[:n_img], reg[:n_img]), sup_feature,
                    ([f[n_img:] for f in feats], cls[n_img:], reg[n_img:]), un_sup_feature)
        from ..autograd import split_batch     # views whose loss gradients are stitched without copies
        halves = [split_batch(p, n_img) for p in total_pred]
        sup_pred = [h[0] for h in halves]
        un_sup_pred = [h[1] for h in halves]
        return sup_pred, sup_feature, un_sup_pred, un_sup_feature

    def train_instance(self, imgs, targets, paths, unlabeled_imgs, unlabeled_imgs_ori, unlabeled_gt, unlabeled_M, ni,
                       pbar=None, callbacks=None):
        """reference signature (ssod_trainer.py:587).  After a few eager steps the whole step is replayed as one HIP graph."""
        if self.use_graph and self._graph_capable() and self._eager_steps >= self.graph_warmup:
            accumulate = 1 if self.fixed_accumulate else max(round(64 / self.batch_size), 1)
            if ni <= self.nw and not self.fixed_accumulate:
                import numpy as np
                accumulate = max(1, np.interp(ni, [0, self.nw], [1, 64 / self.batch_size]).round())
            if self._graph is None:
                from .graph_step import StepGraph
                self._graph = StepGraph(self)
            if accumulate == 1 and self._graph.usable(imgs, targets):
                try:
                    return self._graph.run(imgs, targets, unlabeled_imgs, unlabeled_imgs_ori, unlabeled_M, ni)
                except Exception as e:                 # a capture the runtime rejects (e.g. a collective library that cannot be
                    if self._graph.graph is not None:  # captured) or two captures that replay slower than the eager step: this
                        raise                          # step and all later ones are issued eagerly, loudly (a failure of an
                                                       # already captured graph is a real error)
                    import logging
                    logging.getLogger(__name__).warning("step graph dropped (%s: %s): falling back to eager steps", type(e).__name__, e)
                    self.graph_error = f"{type(e).__name__}: {e}"
                    self.use_graph = False
                    if self.cuda:
                        torch.cuda.synchronize(self.device)
                    if isinstance(self.model, FlatDataParallel):
                        # a capture that failed half-way may have counted gradient-ready hooks or started collectives: the eager
                        # step below starts from clean counters on EVERY rank (all ranks reject the same capture: same code, same signature)
                        self.model.abort_step()
        self._eager_steps += 1
        time_it = self.use_graph and self.cuda and self._eager_steps == self.graph_warmup    # the graph's yardstick (graph_step.py)
        if time_it:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        out = self._train_instance_eager(imgs, targets, paths, unlabeled_imgs, unlabeled_imgs_ori, unlabeled_gt, unlabeled_M, ni,
                                         pbar, callbacks)
        if time_it:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self._eager_events = (e0, e1)
        return out

    def _graph_capable(self):
        """the step graph needs a HIP device (tests override this to drive the fall-back path on CPU ranks)"""
        return self.cuda

    def eager_step_ms(self):
        """HIP-event span of the last eager step before the capture (None when it was not timed)"""
        ev = getattr(self, "_eager_events", None)
        if ev is None:
            return None
        ev[1].synchronize()
        return ev[0].elapsed_time(ev[1])

    def _train_instance_eager(self, imgs, targets, paths, unlabeled_imgs, unlabeled_imgs_ori, unlabeled_gt, unlabeled_M, ni,
                              pbar=None, callbacks=None, sup_table=None):
        n_img = imgs.shape[0]
        height, width = unlabeled_imgs.shape[2], unlabeled_imgs.shape[3]
        # 1+2 teacher forward (ssod_trainer.py:595-599: EMA model, eval, no grad) and pseudo labels (:618).
        # Their result is first needed by the unsupervised loss, AFTER the student forward, so on a GPU
        # they run on a second HIP stream next to it: the teacher's HBM-bound 1x1 layers and partially
        # filled last waves interleave with the student's MFMA-bound kernels.
        side = self._side_stream() if self.cuda and self.overlap_teacher else None
        cur = torch.cuda.current_stream(self.device) if side is not None else None
        if self.cfg.SSOD.pseudo_label_type == 'LabelMatch':         # ssod_trainer.py:616-617
            self.pseudo_label_creator.update(targets, n_img, unlabeled_imgs.shape[0])

        def teacher(start_event=None):
            if side is not None:
                side.wait_stream(cur) if start_event is None else side.wait_event(start_event)
            with torch.no_grad(), (torch.cuda.stream(side) if side is not None else _nullcontext()):
                from .. import ops as _ops
                _ops.SCOPE = "teacher"
                try:
                    (teacher_pred, _), _ = self.ema.ema(unlabeled_imgs_ori, augment=False)
                finally:
                    _ops.SCOPE = None
                if self.teacher_pred_hook is not None:
                    teacher_pred = self.teacher_pred_hook(teacher_pred)
                t9, valid = self.pseudo_label_creator.create_pseudo_label_padded(teacher_pred, unlabeled_M, width, height)
                return t9, valid, valid.any().float()        # has_targets == not invalid_target_shape, as a device flag

        # The teacher stream starts when the student's forward has passed its stride-8 stage (teacher_after = "p3"), not at the start: both
        # networks begin with their large, HBM-bound maps, and running those side by side only makes both slower.  Same-box A/B:
        # r03 / r04 (profiles/r03_teacher_start_ab.txt, r04_knob_combinations_ab.txt, the step at 55-58 ms) chose "p2"; re-tuned at the r06
        # build (profiles/r06_teacher_start_ab.txt): p2 49.85 / start 49.76 / p1 49.78 / p3 49.60 / p4 50.38 ms over 20 steps, p3 -0.1 ms at
        # 100 steps and -0.16 ms in fp16 mode.
        after = self.teacher_after if side is not None else ""
        if after:
            from ..models.backbone.yolov5_backbone import YoloV5BackBone
            inner = self.model.module if isinstance(self.model, FlatDataParallel) else self.model
            if not isinstance(inner.backbone, YoloV5BackBone):
                after = ""                                    # only this backbone marks its stages
        if not after:
            t9, valid, has_targets = teacher()
        # 3 student forward on the concatenated batch (:623-627)
        # the reference concatenates the two batches (:623); here they are packed into one NHWC buffer directly
        same = imgs.shape[1:] == unlabeled_imgs.shape[1:] and imgs.dtype == unlabeled_imgs.dtype
        total_imgs = [imgs, unlabeled_imgs] if same else torch.cat([imgs, unlabeled_imgs.to(imgs.dtype)], 0)
        if after:
            from ..models.backbone import yolov5_backbone as _bb
            if side is not None:
                side.wait_stream(cur)                         # inputs / weights of this step are ready on the main stream
            _bb.STAGE_EVENTS = {}
            try:
                total_pred, total_feature = self.model(total_imgs)
                ev = _bb.STAGE_EVENTS.get(after)
            finally:
                _bb.STAGE_EVENTS = None
            t9, valid, has_targets = teacher(ev)
        else:
            total_pred, total_feature = self.model(total_imgs)
        def join_teacher():
            if side is not None:
                cur.wait_stream(side)
                for t in (t9, valid, has_targets):
                    t.record_stream(cur)
        if not self.join_teacher_late:
            join_teacher()
        sup_pred, sup_feature, un_sup_pred, un_sup_feature = self.split_predict_and_feature(total_pred, total_feature, n_img)
        # 4 losses (:628-649); the zero-weighted domain losses (:631-636) contribute nothing
        if sup_table is not None:                      # graph capture: the padded device-resident target table
            fn = self.compute_loss.ota_loss if getattr(self.compute_loss, 'ota', False) else self.compute_loss.default_loss
            sup_loss, sup_loss_items = fn(sup_pred, None, table=sup_table)
        else:
            sup_loss, sup_loss_items = self.compute_loss(sup_pred, targets.to(self.device))
        if self.cfg.SSOD.with_da_loss:                 # ssod_trainer.py:631-634
            d_loss = self.domain_loss(sup_feature)
            t_loss = self.target_loss(un_sup_feature)
            sup_loss = sup_loss + d_loss * self.da_loss_weights + t_loss * self.da_loss_weights
        if self.RANK != -1:
            sup_loss = sup_loss * self.WORLD_SIZE
        if self.join_teacher_late:                     # the supervised loss does not need the pseudo labels: it runs beside the teacher's NMS tail
            join_teacher()
        un_sup_loss, un_sup_loss_items = self.compute_un_sup_loss(un_sup_pred, t9, valid)
        un_sup_loss = un_sup_loss * has_targets       # reference: zeros(1) when no pseudo label survived (:640-643)
        if self.RANK != -1:
            un_sup_loss = un_sup_loss * self.WORLD_SIZE
        loss = sup_loss + un_sup_loss * self.cfg.SSOD.teacher_loss_weight
        # 5 backward / optimizer / EMAs (:651)
        self.update_optimizer(loss, ni)
        self._last_pseudo = (t9, valid)               # for the adapters' progress-bar statistics (ssod_trainer.py:657-673)
        return dict(sup_loss_items, **un_sup_loss_items)

    def train_with_unlabeled(self, labeled_batches, unlabeled_batches, start_ni=0):
        """ssod_trainer.py:682-697 for iterables of the reference's batch tuples
        (imgs, targets, paths, shapes) and (imgs, targets, paths, shapes, imgs_ori, M_s).  The uint8 image batches are
        staged to the device one step ahead on a copy stream and normalised inside the input pack kernel
        (utils/prefetch.py) instead of the reference's `.to(device).float() / 255.0` on the compute stream."""
        from ..utils.prefetch import DevicePrefetcher
        self.optimizer.zero_grad()
        labeled = DevicePrefetcher(labeled_batches, self.device)
        out = None
        for i, (t_imgs, t_gt, t_paths, _, t_imgs_ori, t_M) in enumerate(DevicePrefetcher(unlabeled_batches, self.device)):
            imgs, targets, paths, _ = next(labeled)
            if t_imgs is None:                     # loader delivered the weak view only: strong view + M_s on the device
                if self._strong_view is None:
                    from ..utils.augment import StrongViewGenerator
                    self._strong_view = StrongViewGenerator(self.cfg.hyp)
                t_imgs, t_M = self._strong_view(t_imgs_ori)
            if not self.cuda:                      # emulator / CPU tests: the plain conversion
                imgs, t_imgs, t_imgs_ori = (x.float() / 255.0 if x.dtype == torch.uint8 else x for x in (imgs, t_imgs, t_imgs_ori))
            out = self.train_instance(imgs, targets, paths, t_imgs, t_imgs_ori, t_gt, t_M, start_ni + i)
        return out
