"""TEST INFRASTRUCTURE -- times the IMPORTED reference (AlibabaResearch/efficientteacher, /root/reference) on the CPU
cores of the BUILD container: one real SSODTrainer.train_instance + update_optimizer (trainer/ssod_trainer.py:587-680,
458-488), YOLOv5l, 640x640, 1 labeled + 1 unlabeled image, synthetic inputs and injected teacher scores as in bench.py.
The reference cannot travel to the GPU box, so this figure is taken here once and stored under profiles/
(bench.py's cpu_baseline on the GPU box is the oracle port, oracle/step.py, which restates this very function).
    python -m oracle.time_reference_step [B] > profiles/r04_reference_vs_port_cpu.json      (B labeled + B unlabeled images, default 2)
Since r04 the same process also times the oracle PORT (oracle/step.py + SGD + EMA, exactly what bench.py's cpu_baseline runs) on the
same cores, same batch, same threads: `port_over_reference` is the ratio bench.py reports as cpu_baseline.reference_ratio.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_loader  # noqa: E402


def time_port(cfg_amd, B, cores):
    """bench.py's cpu_baseline leg (oracle step + SGD + EMA) on this host: seconds per step"""
    import copy
    import bench
    from oracle import model as o_model, step as o_step
    torch.set_num_threads(cores)
    rng = np.random.default_rng(0)
    imgs, targets, u_str, u_ori, M_s = bench.make_batch(rng, B, B, 640, "cpu")
    synth = bench.synth_teacher_scores(cfg_amd, B, 640)
    torch.manual_seed(0)
    student = o_model.Model.from_cfg(cfg_amd).train()
    teacher = copy.deepcopy(student).eval()
    opt = torch.optim.SGD(student.parameters(), lr=0.01, momentum=0.937, nesterov=True)

    def step():
        opt.zero_grad()
        o_step.ssod_step(student, teacher, imgs, targets, u_str, u_ori, M_s, cfg_amd, synth_scores=synth)
        opt.step()
        with torch.no_grad():
            for v, m in zip(teacher.state_dict().values(), student.state_dict().values()):
                if v.dtype.is_floating_point:
                    v.mul_(0.9999).add_(m, alpha=1e-4)
    step()
    n, t0 = 0, time.time()
    while n < 2 or (time.time() - t0 < 20 and n < 6):
        step(); n += 1
    return (time.time() - t0) / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    import bench as _bench
    cfg_amd = _bench.load_cfg(_bench.WORKLOADS["v5l-ssod"], 2 * B, ["SSOD.fixed_accumulate", True])
    port_s = time_port(cfg_amd, B, os.cpu_count() or 1)
    ref_loader.load()
    from torch.cuda import amp
    from models.detector.yolo_ssod import Model
    from models.loss.loss import ComputeLoss, DomainLoss, TargetLoss
    from models.loss.ssod.ssod_loss import ComputeStudentMatchLoss
    from trainer.ssod_trainer import SSODTrainer
    from utils.self_supervised_utils import FairPseudoLabel
    from utils.torch_utils import ModelEMA, SemiSupModelEMA
    import bench
    cfg = ref_loader.get_cfg("configs/ssod/coco-standard/yolov5l_coco_ssod_10_percent.yaml",
                             ["Dataset.batch_size", 2 * B, "SSOD.fixed_accumulate", True, "device", "cpu"])
    cfg.freeze()
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = Model(cfg).train()
    t = object.__new__(SSODTrainer)
    t.cfg = cfg; t.model = m; t.model_type = 'yolov5'; t.cuda = False; t.device = torch.device('cpu')
    t.RANK = 1; t.WORLD_SIZE = 1; t.extra_teacher_models = []          # RANK 1: skips the rank-0 logging block only
    t.epochs = cfg.epochs; t.epoch = 0; t.batch_size = cfg.Dataset.batch_size
    t.ema = ModelEMA(m)
    t.semi_ema = SemiSupModelEMA(t.ema.ema, cfg.SSOD.ema_rate)
    t.pseudo_label_creator = FairPseudoLabel(cfg)
    t.compute_loss = ComputeLoss(m, cfg)
    t.compute_un_sup_loss = ComputeStudentMatchLoss(m, cfg)
    t.domain_loss = DomainLoss(); t.target_loss = TargetLoss()
    t.da_loss_weights = cfg.SSOD.da_loss_weights
    t.fixed_accumulate = cfg.SSOD.fixed_accumulate
    t.scaler = amp.GradScaler(enabled=False)
    t.accumulate = 1
    g_bnw, g_w, g_b = [], [], []
    for v in m.modules():
        if hasattr(v, 'bias') and isinstance(v.bias, torch.nn.Parameter):
            g_b.append(v.bias)
        if isinstance(v, torch.nn.BatchNorm2d):
            g_bnw.append(v.weight)
        elif hasattr(v, 'weight') and isinstance(v.weight, torch.nn.Parameter):
            g_w.append(v.weight)
    t.optimizer = torch.optim.SGD(g_b, lr=cfg.hyp.lr0, momentum=cfg.hyp.momentum, nesterov=True)
    t.optimizer.add_param_group({'params': g_w, 'weight_decay': cfg.hyp.weight_decay * 2 / 64})
    t.optimizer.add_param_group({'params': g_bnw})
    t.lf = lambda x: 1.0
    t.nw = -1; t.warmup_bias_lr = cfg.hyp.warmup_bias_lr; t.warmup_momentum = cfg.hyp.warmup_momentum
    t.momentum = cfg.hyp.momentum; t.last_opt_step = -1
    rng = np.random.default_rng(0)
    imgs, targets, u_str, u_ori, M_s = bench.make_batch(rng, B, B, 640, "cpu")
    synth = torch.rand(B, 25200, 81) ** torch.cat((torch.full((1,), 16.0), torch.full((80,), 4.0)))
    ema_fwd = t.ema.ema.forward

    def fwd(x, augment=False):                      # inject the synthetic teacher scores (bench.py's teacher_pred_hook)
        (tp, tr_out), feat = ema_fwd(x, augment=augment)
        tp[..., 4:] = synth
        return (tp, tr_out), feat
    t.ema.ema.forward = fwd

    def step(ni):
        t.train_instance(imgs, targets, None, u_str, u_ori, None, M_s, ni, None, None)

    t.optimizer.zero_grad()
    t0 = time.time(); step(0); warm = time.time() - t0
    n, t0 = 0, time.time()
    while n < 2 or (time.time() - t0 < 20 and n < 6):
        step(1 + n); n += 1
    dt = (time.time() - t0) / n
    print(json.dumps(dict(what=f"imported reference SSODTrainer.train_instance + update_optimizer, YOLOv5l 640x640, {B} labeled + {B} unlabeled "
                               "images, fp32 CPU (torchvision.ops.nms stubbed by oracle/nms.py, see oracle/ref_loader.py) -- and the oracle "
                               "port (oracle/step.py + SGD + EMA = bench.py's cpu_baseline leg) on the same cores and batch",
                          where="build container", cores=cores, torch=torch.__version__, steps=n, first_step_s=warm, s_per_step=dt,
                          images_per_s=2.0 * B / dt, port_s_per_step=port_s, port_images_per_s=2.0 * B / port_s,
                          port_over_reference=dt / port_s)))


if __name__ == "__main__":
    main()
