"""TEST INFRASTRUCTURE -- times the IMPORTED reference (AlibabaResearch/efficientteacher) on CPU cores: real
SSODTrainer.train_instance + update_optimizer steps (trainer/ssod_trainer.py:587-680, 458-488), YOLOv5l, 640x640, B labeled +
B unlabeled images, synthetic inputs and injected teacher scores as in bench.py.

Where the reference comes from (oracle/ref_loader.py, ET_REFERENCE):
  * /root/reference in the build container (the live, read-only tree), or
  * oracle/_ref/ -- the byte-compiled image of exactly the reference modules this step imports, produced by
    oracle/make_ref.py (git-ignored, it ships to the GPU box with the push like libet_hip.so): bench.py's `cpu_baseline` leg
    runs this script against it ON THE GPU BOX'S HOST CORES, in the same bench run (`cpu_baseline.kind == "reference"`).

    python -m oracle.time_reference_step [B] [--cores N] [--seconds S] [--port] [--tiny]
`--port` also times the oracle PORT (oracle/step.py + SGD + EMA) in the same process on the same cores; `--tiny` runs a reduced
width / 64-pixel step (make_ref.py uses it to discover the imported modules).  Prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import ref_loader  # noqa: E402

# this package's canonical copy of the recipe's resolved values (same keys as the reference's yaml of that name)
SSOD_YAML = os.path.join(ROOT, "efficientteacher_amd", "configs", "ssod", "coco-standard", "yolov5l_coco_ssod_10_percent.yaml")


def time_port(cfg_amd, B, cores, seconds):
    """bench.py's port leg (oracle step + SGD + EMA) on this host: seconds per step"""
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
    while n < 2 or (time.time() - t0 < seconds and n < 8):
        step(); n += 1
    return (time.time() - t0) / n


def build_reference_trainer(cfg):
    """An SSODTrainer of the imported reference around a fresh Model(cfg), assembled field by field (its __init__ wants datasets
    and a run directory): exactly the attributes train_instance / update_optimizer read."""
    from torch.cuda import amp
    from models.detector.yolo_ssod import Model
    from models.loss.loss import ComputeLoss, DomainLoss, TargetLoss
    from models.loss.ssod.ssod_loss import ComputeStudentMatchLoss
    from trainer.ssod_trainer import SSODTrainer
    from utils.self_supervised_utils import FairPseudoLabel
    from utils.torch_utils import ModelEMA, SemiSupModelEMA
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
    return t


def reference_step_fn(B, S, tiny):
    """-> (step(ni), cfg): one imported-reference train_instance + update_optimizer on a fixed synthetic batch"""
    import bench
    ref_loader.load()
    opts = ["Dataset.batch_size", 2 * B, "SSOD.fixed_accumulate", True, "device", "cpu"]
    if tiny:
        opts += ["Model.width_multiple", 0.125, "Model.depth_multiple", 0.33]
    cfg = ref_loader.get_cfg(None, ())
    cfg.merge_from_file(SSOD_YAML)
    cfg.merge_from_list(opts)
    cfg.freeze()
    torch.manual_seed(0)
    t = build_reference_trainer(cfg)
    rng = np.random.default_rng(0)
    imgs, targets, u_str, u_ori, M_s = bench.make_batch(rng, B, B, S, "cpu")
    na = 3 * sum((S // s) ** 2 for s in (8, 16, 32))
    synth = torch.rand(B, na, 81) ** torch.cat((torch.full((1,), 16.0), torch.full((80,), 4.0)))
    ema_fwd = t.ema.ema.forward

    def fwd(x, augment=False):                      # inject the synthetic teacher scores (bench.py's teacher_pred_hook)
        (tp, tr_out), feat = ema_fwd(x, augment=augment)
        tp[..., 4:] = synth
        return (tp, tr_out), feat
    t.ema.ema.forward = fwd
    t.optimizer.zero_grad()

    def step(ni):
        t.train_instance(imgs, targets, None, u_str, u_ori, None, M_s, ni, None, None)
    return step, cfg


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("B", nargs="?", type=int, default=2)
    ap.add_argument("--cores", type=int, default=0, help="torch threads (default: min(host cores, 32))")
    ap.add_argument("--seconds", type=float, default=20.0, help="bound on the timed work")
    ap.add_argument("--port", action="store_true", help="also time the oracle port in this process")
    ap.add_argument("--tiny", action="store_true", help="reduced width, 64-pixel images (module discovery)")
    a = ap.parse_args(argv)
    B = a.B
    # beyond ~32 threads the many small layers of a small batch only add oversubscription (bench.py: measured on the 256-core host)
    cores = a.cores or min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    S = 64 if a.tiny else 640
    out = dict(where=ref_loader.REF, cores=cores, torch=torch.__version__)
    if a.port:
        import bench as _bench
        cfg_amd = _bench.load_cfg(_bench.WORKLOADS["v5l-ssod"], 2 * B, ["SSOD.fixed_accumulate", True])
        port_s = time_port(cfg_amd, B, cores, a.seconds)
        out.update(port_s_per_step=port_s, port_images_per_s=2.0 * B / port_s)
    step, _ = reference_step_fn(B, S, a.tiny)
    t0 = time.time(); step(0); warm = time.time() - t0
    n, t0 = 0, time.time()
    while n < 2 or (time.time() - t0 < a.seconds and n < 8):
        step(1 + n); n += 1
    dt = (time.time() - t0) / n
    out.update(what=f"imported reference SSODTrainer.train_instance + update_optimizer, YOLOv5l {S}x{S}, {B} labeled + {B} unlabeled "
                    "images, fp32 CPU (torchvision.ops.nms stubbed by oracle/nms.py, see oracle/ref_loader.py)",
               steps=n, first_step_s=warm, s_per_step=dt, images_per_s=2.0 * B / dt)
    if a.port:
        out["port_over_reference"] = dt / out["port_s_per_step"]
    # every reference package module must have come from ET_REFERENCE (a shipped image that silently fell back to another tree
    # would not be the thing named in `where`)
    ref_real = os.path.realpath(ref_loader.REF) + os.sep
    stray = sorted(n for n, m in list(sys.modules.items())
                   if n.split(".")[0] in ("models", "utils", "trainer", "configs") and getattr(m, "__file__", None)
                   and not os.path.realpath(m.__file__).startswith(ref_real))
    assert not stray, f"reference modules imported from outside {ref_loader.REF}: {stray}"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
