"""bench.py -- images/sec of one Efficient-Teacher training step on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload v5l-ssod | v5s-sup | v8-sup]
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Default workload ``v5l-ssod`` (BASELINE.json metric): SSOD images/sec (teacher + student step), YOLOv5l, 640 px.
One step = SSODTrainer.train_instance on one batch of synthetic input already resident in HBM:
EMA-teacher inference on the unlabeled weak view -> NMS + pseudo-label transform -> student forward on
cat(labeled, unlabeled) -> ComputeLoss + ComputeStudentMatchLoss -> backward (+ RCCL gradient
all-reduce for N > 1) -> SGD step + ModelEMA + semi-EMA update (optimizer every step:
SSOD.fixed_accumulate).  Per-GPU batch: N = 1, 2, 4: 32 labeled + 32 unlabeled images per rank (BASELINE configs[2] at
N = 1, weak scaling); N = 8: 16 + 16 per rank = global 128 + 128, which IS BASELINE configs[3] -- the 32 + 32-per-rank
weak-scaling point of the same 8 ranks is measured right after it and reported as ``weak_scaling_point``.
value = N * images per rank per step * K / max-over-ranks time.

Other workloads (the other single-GPU configs of BASELINE.json, same contract, same fields):
  v5s-sup : configs[1], YOLOv5s supervised, bf16, batch 64 per GPU: Trainer.train_step (trainer/trainer.py:406-443)
  v8-sup  : the kernels of configs[4]: YOLOv8 (C2f / decoupled DFL head, width = depth = 1.0) supervised step with
            TaskAlignedAssigner + DFL loss, batch 32 per GPU (the reference has no runnable v8 SSOD step: SURVEY.md 8 a-14)

Extra objects on the JSON line:
  roofline            : the dominant kernel (bf16 MFMA implicit-GEMM conv), algorithmic FLOPs per launch divided by its
                        HIP-event-measured average launch duration inside the timed region.
  kernel_ms_by_family : HIP-event time budget of ONE step by kernel family and stream (one extra instrumented step right
                        after the timed region, same overlap mode), so that the line itself shows where the step goes.
  cpu_baseline        : the plain-torch CPU port of the same step (oracle/) timed on the host cores, rank 0 at N = 1 only,
                        on a bounded sample;  parity_check: fp32-mode and bf16-mode HIP steps against that oracle step.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG_DIR = os.path.join(ROOT, "efficientteacher_amd", "configs")
YAML = os.path.join(CFG_DIR, "ssod", "coco-standard", "yolov5l_coco_ssod_10_percent.yaml")
F_IMG = 111.52e9          # conv FLOPs / image forward, YOLOv5l SSOD model
F_NETD_IMG = 2.52805e9    # ... of which the six netD convs (256->256 @80^2, 512->512 @40^2, 1024->1024 @20^2 and their 2-channel heads):
                          # with SSOD.with_da_loss False the step adds d_loss * 0 (ssod_trainer.py:633-636) and the netD BACKWARD is
                          # never executed (models/detector/yolo_ssod.py:44-55) -- its 2 F_netD per image are not charged as executed work
PEAK_BF16 = 2.5e15        # dense bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_HBM = 8.0e12         # HBM3E spec, MI355X_MICROARCH.md; ACHIEVABLE_HBM: the best streaming pass measured on this part (BN apply pass,
ACHIEVABLE_HBM = 6.3e12   # profiles/r03_*: 6.2-6.3 TB/s)
# Algorithmic HBM bytes of one YOLOv5l SSOD step, SURVEY.md 8(d): every conv reads its input and writes its output once in bf16
# (358.6 MB per image and pass; passes = teacher forward, student forward, dgrad, wgrad), weights 95.9 MB per pass (+ the fp32
# gradient), train-mode BatchNorm + SiLU 4 B/element forward and 10 B/element backward over 78.5 M elements per student image,
# NMS scan, EMA (two of them) and SGD over 47.9 M parameters.
ALG_ACT_BYTES_IMG_PASS = 358.6e6
ALG_BN_ELEMS_IMG = 78.5e6


def algorithmic_bytes_per_step(Bl, Bu):
    conv = ALG_ACT_BYTES_IMG_PASS * (Bu + 3 * (Bl + Bu)) + 4 * 95.9e6 + 47.9e6 * 4
    bn = ALG_BN_ELEMS_IMG * (Bl + Bu) * (4 + 10)
    other = Bu * 25200 * 85 * 4 + 3 * (Bl + Bu) * 25200 * 6 * 4 + 48.0e6 * 4 * 3 * 2 + 47.9e6 * 4 * 5
    return dict(conv_activations_and_weights=conv, batchnorm_silu=bn, nms_loss_ema_sgd=other, total=conv + bn + other)

WORKLOADS = {
    "v5l-ssod": dict(kind="ssod", yaml=YAML, merge=[], per_rank=32,
                     metric="SSOD images/sec (teacher+student step) YOLOv5l 640px",
                     name="YOLOv5l Efficient-Teacher SSOD"),
    "v8-ssod": dict(kind="ssod", yaml=os.path.join(CFG_DIR, "ssod", "coco-standard", "yolov8_coco_ssod_10_percent.yaml"), merge=[], per_rank=32,
                    metric="SSOD images/sec (teacher+student step) YOLOv8 (width = depth = 1.0) 640px, TAL losses (EXTENSION: no reference step)",
                    name="YOLOv8 anchor-free SSOD (BASELINE configs[4]; the unsupervised TAL loss is an extension, the reference has no v8 SSOD step)"),
    "v5s-sup": dict(kind="sup", yaml=os.path.join(CFG_DIR, "sup", "public", "yolov5s_coco.yaml"), merge=[], per_rank=64,
                    metric="supervised images/sec (train step) YOLOv5s 640px bf16", name="YOLOv5s supervised (BASELINE configs[1])"),
    "v8-sup": dict(kind="sup", yaml=os.path.join(CFG_DIR, "sup", "public", "yolov8m_coco.yaml"),
                   merge=["Model.width_multiple", 1.0, "Model.depth_multiple", 1.0], per_rank=32,
                   metric="supervised images/sec (train step) YOLOv8 (width = depth = 1.0) 640px bf16, TaskAlignedAssigner + DFL loss",
                   name="YOLOv8 anchor-free head, supervised TAL step (the kernels of BASELINE configs[4])"),
}


def synth_targets(rng, B):
    rows = []
    for b in range(B):
        n = int(rng.integers(1, 17))
        xy = rng.uniform(0.1, 0.9, (n, 2))
        wh = np.exp(rng.uniform(np.log(0.02), np.log(0.6), (n, 2)))
        wh = np.minimum(wh, 2 * np.minimum(xy, 1 - xy))
        rows.append(np.concatenate((np.full((n, 1), b), rng.integers(0, 80, (n, 1)), xy, wh), 1))
    return torch.from_numpy(np.concatenate(rows, 0).astype(np.float32))


def make_batch(rng, Bl, Bu, S, device):
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    imgs = torch.randint(0, 256, (Bl, 3, S, S), generator=g, dtype=torch.uint8)
    u_ori = torch.randint(0, 256, (Bu, 3, S, S), generator=g, dtype=torch.uint8)
    M_s = torch.zeros(Bu, 13, dtype=torch.float64)
    for i in range(Bu):          # fixed affine: scale 0.8, translate 0.1*S, lr flip on odd images
        s = 0.8
        M_s[i] = torch.tensor([i, s, 0, 0.1 * S, 0, s, 0.1 * S, 0, 0, 1, s, 0, i % 2], dtype=torch.float64)
    f = lambda t: (t.to(device).float() / 255.0)
    return f(imgs), synth_targets(rng, Bl).to(device), f(u_ori), f(u_ori), M_s.to(device)


def synth_teacher_scores(cfg, B, S, generator=None):
    """a random-init teacher scores nothing above conf 0.1, so the objectness / class columns of the teacher output
    are replaced: obj = U^16, cls = U^4 on the anchor-based head; on the anchor-free head the objectness column is the constant 1
    the head itself emits and the class scores are U^16 (the confidence of a detection is then obj * cls in both cases)"""
    if cfg.Loss.type == 'ComputeTalLoss':
        A = (S // 8) ** 2 + (S // 16) ** 2 + (S // 32) ** 2
        sc = torch.rand(B, A, 81, generator=generator) ** torch.cat((torch.full((1,), 1.0), torch.full((80,), 16.0)))
        sc[..., 0] = 1.0
        return sc
    A = 3 * ((S // 8) ** 2 + (S // 16) ** 2 + (S // 32) ** 2)
    return torch.rand(B, A, 81, generator=generator) ** torch.cat((torch.full((1,), 16.0), torch.full((80,), 4.0)))


def load_cfg(wl, batch_size, extra=()):
    from efficientteacher_amd.configs import get_cfg
    cfg = get_cfg()
    cfg.merge_from_file(wl["yaml"])
    cfg.merge_from_list(list(wl["merge"]) + ["Dataset.batch_size", batch_size] + list(extra))
    cfg.freeze()
    return cfg


DTYPES = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}


def build_trainer(device, rank, world, local_rank, per_rank, wl=None, dtype="bf16"):
    wl = wl or WORKLOADS["v5l-ssod"]
    torch.manual_seed(0)
    lr, rk = (local_rank if world > 1 else -1), (rank if world > 1 else -1)
    if os.environ.get("ET_DP_SINGLE_RANK", "0") == "1" and world == 1 and dist.is_initialized():
        lr, rk = local_rank, rank                        # single-GPU box: run the data-parallel code path over a 1-rank RCCL group
    if wl["kind"] == "ssod":
        from efficientteacher_amd.trainer import SSODTrainer
        cfg = load_cfg(wl, per_rank * world, ["SSOD.fixed_accumulate", True])
        return cfg, SSODTrainer(cfg, device, None, lr, rk, world, nb=1000, amp_dtype=DTYPES[dtype])
    from efficientteacher_amd.trainer import Trainer
    cfg = load_cfg(wl, max(64, per_rank * world))        # accumulate = max(round(64 / batch), 1) = 1: optimizer every step
    return cfg, Trainer(cfg, device, None, lr, rk, world, nb=1000, amp_dtype=DTYPES[dtype])


# ---- CPU baseline + parity ---------------------------------------------------------------------------------------------
def _hip_ssod_losses(cfg, device, dtype, batch, synth, deterministic=False, bn_gamma=None, want_grads=False):
    """one HIP SSODTrainer.train_instance in compute dtype `dtype` from seed-0 weights on `batch`; -> (items, teacher_pred, state_dict
    [, conv-weight gradients]).  deterministic: Model.set_deterministic (reproducible BatchNorm statistics in the 16-bit modes).
    bn_gamma: every BatchNorm weight set to this value first (the conditioned init of tests/test_step_fullsize.py); want_grads: the
    parameter gradients, recovered from the first SGD-nesterov update dp = -lr (1 + m) (g + wd p)."""
    from efficientteacher_amd.trainer import SSODTrainer
    from efficientteacher_amd.utils.torch_utils import ModelEMA
    imgs, targets, u_str, u_ori, M_s = batch
    torch.manual_seed(0)
    tr = SSODTrainer(cfg, device, None, -1, -1, 1, nb=1000)
    if bn_gamma is not None:
        with torch.no_grad():
            for m in tr.model.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.weight.fill_(float(bn_gamma))
    if dtype != torch.bfloat16 or deterministic or bn_gamma is not None:
        tr.model._deterministic = bool(deterministic)
        tr.model.set_compute_dtype(dtype)
        tr.build_optimizer(cfg)
        if dtype == torch.float16 and want_grads:
            from efficientteacher_amd.optim import DeviceGradScaler
            tr.scaler = DeviceGradScaler(device, init_scale=256.0)      # GradScaler's initial 65536 skips the first steps by design
        tr.ema = ModelEMA(tr.model)
        tr.semi_ema = None
    assert tr.model.flat_state().deterministic == bool(deterministic)
    sd = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
    p0 = {k: v.detach().clone() for k, v in tr.model.named_parameters()} if want_grads else None
    cap = {}

    def hook(tp):
        tp[..., 4:] = synth.to(device)
        cap["tp"] = tp.detach().clone()
        return tp
    tr.teacher_pred_hook = hook
    items = tr.train_instance(imgs.to(device), targets.to(device), None, u_str.to(device), u_ori.to(device), None,
                              M_s.to(device), 2000)
    items = {k: float(v) for k, v in items.items()}
    tp = cap["tp"]
    grads = None
    if want_grads:
        grads = {}
        groups = {id(p): g for g in tr.optimizer.param_groups for p in g["params"]}
        for n, p in tr.model.named_parameters():
            g = groups.get(id(p))
            if g is not None and p.dim() == 4:
                lr, m, wd = float(g["lr"]), float(g["momentum"]), float(g["weight_decay"])
                grads[n] = (-(p.detach() - p0[n]) / (lr * (1.0 + m)) - wd * p0[n]).cpu()
    del tr
    torch.cuda.empty_cache()
    return (items, tp, sd, grads) if want_grads else (items, tp, sd)


def reference_timing(Bl, Bu, S, seconds=20.0):
    """The IMPORTED reference (SSODTrainer.train_instance + update_optimizer, /root/reference's own code) timed on THIS host's
    cores in THIS run: oracle/time_reference_step.py in a subprocess (its import shims patch torch and sys.modules), against
    ET_REFERENCE if set, else the shipped byte-compiled image oracle/_ref/ (oracle/make_ref.py: git-ignored, built by
    __graft_entry__.build() in the build container, travels with the push), else the live /root/reference.  None when no
    reference is reachable (a checkout without the image): the caller then reports the port, `kind: "port"`."""
    import subprocess
    ref_root = os.environ.get("ET_REFERENCE")
    if not ref_root:
        for cand in (os.path.join(ROOT, "oracle", "_ref"), "/root/reference"):
            if os.path.isdir(os.path.join(cand, "models")):
                ref_root = cand
                break
    if not ref_root or Bl != Bu:
        return None
    cores = min(os.cpu_count() or 1, 32)
    try:
        # the GPU is hidden from this child: the reference hard-codes `.cuda()` in its losses (models/loss/loss.py:392,418) and the
        # import shims (oracle/ref_loader.py) turn that into the identity only where torch sees no device -- this leg is the CPU path
        env = dict(os.environ, ET_REFERENCE=ref_root, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
        p = subprocess.run([sys.executable, "-m", "oracle.time_reference_step", str(Bl), "--cores", str(cores), "--seconds", str(seconds)],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        if p.returncode != 0 or not p.stdout.strip():
            return dict(error=f"oracle.time_reference_step rc {p.returncode}: {p.stderr.strip()[-600:]}")
        r = json.loads(p.stdout.strip().splitlines()[-1])
        mf = os.path.join(ref_root, "MANIFEST.json")      # oracle/make_ref.py: module -> sha256 of the source it was compiled from
        mf_hash = None
        if os.path.isfile(mf):
            import hashlib
            mf_hash = hashlib.sha256(open(mf, "rb").read()).hexdigest()
        return dict(value=r["images_per_s"], cores=r["cores"], steps=r["steps"], s_per_step=r["s_per_step"],
                    where=os.path.relpath(ref_root, ROOT) if ref_root.startswith(ROOT) else ref_root, manifest_sha256=mf_hash)
    except Exception as e:
        return dict(error=f"{type(e).__name__}: {e}"[:400])


def cpu_baseline_ssod(cfg, device, seconds=25.0, Bl=2, Bu=2):
    """The oracle (oracle/step.py: plain-torch CPU restatement of trainer/ssod_trainer.py:587-680, `kind: "port"`) timed on
    the host cores for a bounded number of Bl + Bu image steps -- and, on the way, the PARITY CHECK of the benchmarked
    configuration: the very first oracle step and one HIP trainer step in EACH compute mode (fp32 parity mode, bf16
    performance mode) start from the same weights and see the same images / targets / M_s / injected teacher scores; their
    loss terms, pseudo-label sets and NMS decisions are compared (what tests/test_step_fullsize.py asserts at 1 + 1 images,
    reported here at Bl + Bu next to the throughput)."""
    import copy
    from efficientteacher_amd.utils.general import nms_ssod_padded
    from oracle import model as o_model, nms as o_nms, step as o_step
    # 32 threads: beyond that the many small layers of a small batch only add oversubscription
    # (measured on the 256-core MI355X host: 256 threads are ~100x slower than 8)
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    rng = np.random.default_rng(0)
    S = cfg.Dataset.img_size
    c2 = cfg.clone(); c2.defrost(); c2.merge_from_list(["Dataset.batch_size", Bl + Bu]); c2.freeze()
    batch = make_batch(rng, Bl, Bu, S, "cpu")
    imgs, targets, u_str, u_ori, M_s = batch
    synth = synth_teacher_scores(cfg, Bu, S)
    v8 = cfg.Loss.type == 'ComputeTalLoss'
    hip = {}
    # fp32 parity mode; the 16-bit modes on the REPRODUCIBLE BatchNorm path (Model.set_deterministic: the mode their tight bounds are
    # stated in) and once more in the default mode of the timed region (sharded fp32 accumulators, atomic order in the last bits)
    for name, dt, det in (("fp32", torch.float32, False), ("bf16", torch.bfloat16, True), ("fp16", torch.float16, True),
                          ("bf16_default_mode", torch.bfloat16, False), ("fp16_default_mode", torch.float16, False)):
        hip[name] = _hip_ssod_losses(c2, device, dt, batch, synth, deterministic=det)
    if not v8:                               # reproducibility itself: the fp16 deterministic step a second time
        hip["fp16_again"] = _hip_ssod_losses(c2, device, torch.float16, batch, synth, deterministic=True)
    if v8:
        from oracle import v8 as o_v8
        student = o_v8.Model.from_cfg(cfg)
        miss = student.load_state_dict(hip["fp32"][2], strict=False)       # the oracle v8 model has no netD branch
        assert not miss.missing_keys and all(k.startswith("det_") for k in miss.unexpected_keys), miss
    else:
        student = o_model.Model.from_cfg(cfg)
        student.load_state_dict(hip["fp32"][2], strict=True)
    student.train()
    teacher = copy.deepcopy(student).eval()
    opt = torch.optim.SGD(student.parameters(), lr=0.01, momentum=0.937, nesterov=True)

    def step():
        opt.zero_grad()
        r = (o_step.ssod_step_v8 if v8 else o_step.ssod_step)(student, teacher, imgs, targets, u_str, u_ori, M_s, cfg, synth_scores=synth)
        opt.step()
        with torch.no_grad():
            for v, m in zip(teacher.state_dict().values(), student.state_dict().values()):
                if v.dtype.is_floating_point:
                    v.mul_(0.9999).add_(m, alpha=1e-4)
        return r

    t0 = time.time()
    ref = step()                             # warm-up (allocator, oneDNN primitive caches) == the parity reference
    warm = time.time() - t0
    want = {**{k: ref["sup_items"][k] for k in (("loss_iou", "loss_dfl", "loss_cls") if v8 else ("box", "obj", "cls"))}, **ref["un_items"]}
    TOL = {"fp32": 1e-4, "bf16": 5e-2, "fp16": 5e-3, "bf16_default_mode": 5e-2, "fp16_default_mode": 1e-2}
    parity = dict(against=(f"oracle/step.py::ssod_step_v8 (EXTENSION: written specification, parity unpinned by construction), {Bl}+{Bu} images" if v8 else
                           f"oracle/step.py (fp32 CPU restatement of the reference step), same weights and inputs, {Bl}+{Bu} images"),
                  tolerance="loss terms, relative: fp32 mode 1e-4; bf16 5e-2 (bf16 storage, fp32 accumulation); fp16 (the reference's autocast "
                            "dtype) 5e-3 on the reproducible BatchNorm path (Model.set_deterministic / cfg.Model.deterministic_bn) and 1e-2 in the "
                            "default mode of the timed region (`*_default_mode`: sharded fp32 accumulators, whose last bits depend on atomic order "
                            "-- 0.8e-3 ... 6e-3 by run); NMS kept indices bit-exact on identical decoded inputs in every mode "
                            "(tests/test_step_fullsize.py)",
                  tolerances=TOL)
    for name in TOL:
        items, tp, _ = hip[name]
        rel = {k: abs(items[k] - v) / max(abs(v), 1e-12) for k, v in want.items()}
        dets, counts, keep, _ = nms_ssod_padded(tp, cfg.SSOD.nms_conf_thres, cfg.SSOD.nms_iou_thres)
        rd, rk = o_nms.non_max_suppression_ssod(tp.cpu().numpy(), cfg.SSOD.nms_conf_thres, cfg.SSOD.nms_iou_thres)
        keep_ok = all(int(counts[i]) == rk[i].shape[0] and np.array_equal(keep[i, :int(counts[i])].cpu().numpy(), rk[i])
                      for i in range(Bu))
        parity[name] = dict(loss_rel_dev={k: round(v, 7) for k, v in rel.items()}, max_loss_rel_dev=max(rel.values()),
                            nms_keep_indices_bit_exact=bool(keep_ok),
                            n_pseudo_labels=[int(counts.sum()), int(sum(k.shape[0] for k in rk))],
                            batchnorm_statistics=("fp64 finalize of partial rows" if name == "fp32" else
                                                  "sharded fp32 atomics (default)" if name.endswith("default_mode") else
                                                  "reproducible partial rows (deterministic switch)"),
                            within_tolerance=bool(max(rel.values()) <= TOL[name] and keep_ok))
    if "fp16_again" in hip:
        a, b = hip["fp16"], hip["fp16_again"]
        # loss items are collected with fp32 atomics over workgroups (loss.hip): equal to the last bits of that sum; the decoded
        # teacher output (every conv / BatchNorm of an eval forward) bit for bit
        parity["fp16"]["second_run_max_rel_diff_of_loss_items"] = max(abs(a[0][k] - b[0][k]) / max(abs(a[0][k]), 1e-12) for k in a[0])
        parity["fp16"]["second_run_teacher_output_bit_equal"] = bool(torch.equal(a[1], b[1]))
    if not v8:
        # weight gradients of the fp16 mode against the fp32 ORACLE at the conditioned init (every BatchNorm weight 0.3: DESIGN.md 3
        # explains why no reduced-precision path can be bounded at the default init of this network): cosine per conv tensor
        try:
            items_g, _, sd_g, grads = _hip_ssod_losses(c2, device, torch.float16, batch, synth, deterministic=True, bn_gamma=0.3, want_grads=True)
            st2 = o_model.Model.from_cfg(cfg)
            st2.load_state_dict(sd_g, strict=True)
            st2.train()
            te2 = copy.deepcopy(st2).eval()
            o_step.ssod_step(st2, te2, imgs, targets, u_str, u_ori, M_s, cfg, synth_scores=synth)
            cos = {}
            for n, p in st2.named_parameters():
                g = grads.get(n)
                if g is None or p.grad is None or p.dim() != 4 or float(p.grad.norm()) == 0.0:
                    continue
                cos[n] = torch.nn.functional.cosine_similarity(g.flatten().double(), p.grad.flatten().double(), 0).item()
            vals = sorted(cos.values())
            parity["fp16"]["conv_grad_cosine_vs_fp32_oracle"] = dict(
                at="conditioned init (BatchNorm weights 0.3), loss scale 256, reproducible BatchNorm path", tensors=len(vals),
                min=vals[0], median=vals[len(vals) // 2], worst_tensor=min(cos, key=cos.get), bound=0.995,
                within_bound=bool(vals[0] >= 0.995),
                note="bound = tests/test_step_fullsize.py::test_yolov5l_640_ssod_step_fp16_gradients_vs_oracle; reproducible to 1e-4 for one build "
                     "(fp32 atomics of the weight-gradient split-K), but a draw from 0.9962 ... 0.9992 ACROSS builds: a one-ulp regrouping of the "
                     "BatchNorm partial sums re-rolls the fp16 roundings downstream and moves every tensor's cosine together "
                     "(profiles/r06_fp16_grad_cosine_sensitivity.txt); the bf16 mode's bound at the same point is 0.95")
            del st2, te2, grads
        except Exception as e:              # the gradient leg must never cost the line its throughput number
            parity["fp16"]["conv_grad_cosine_vs_fp32_oracle"] = dict(error=f"{type(e).__name__}: {e}"[:300])
    hip.clear()
    torch.cuda.empty_cache()
    t0, n = time.time(), 0
    if warm < seconds:                       # bounded: ~`seconds` of timed CPU work, at least one step
        while n < 1 or (time.time() - t0 + warm < seconds and n < 8):
            step(); n += 1
        dt = (time.time() - t0) / n
    else:
        n, dt = 1, warm
    base = dict(value=(Bl + Bu) / dt, unit="images/s", cores=cores, kind="port",
                sample=f"{'YOLOv8' if v8 else 'YOLOv5l'} SSOD step, {Bl} labeled + {Bu} unlabeled {S}x{S}, {n} steps, plain-torch fp32 CPU port "
                       f"(oracle/step.py; all {ref['t9'].shape[0]} pseudo labels)")
    if not v8:
        ref_t = reference_timing(Bl, Bu, S)
        if ref_t and "value" in ref_t:
            # north_star: "the reference's CPU path timed on the host cores of the same box in the same run"
            port = base
            base = dict(value=ref_t["value"], unit="images/s", cores=ref_t["cores"], kind="reference",
                        sample=f"the IMPORTED reference's SSODTrainer.train_instance + update_optimizer (trainer/ssod_trainer.py:587-680, 458-488), "
                               f"YOLOv5l, {Bl} labeled + {Bu} unlabeled {S}x{S}, {ref_t['steps']} steps, fp32 CPU, from {ref_t['where']} "
                               "(oracle/make_ref.py; torchvision.ops.nms stubbed by oracle/nms.py)",
                        port=port, port_over_reference=port["value"] / ref_t["value"],
                        reference_image=dict(present=True, where=ref_t["where"], manifest_sha256=ref_t.get("manifest_sha256")))
        elif ref_t:
            base["reference_error"] = ref_t["error"]
            base["reference_image"] = dict(present=True, error=True)
        else:
            # a checkout without the git-ignored oracle/_ref (built by __graft_entry__.build() where /root/reference exists): the
            # baseline below is the oracle PORT -- said here so that the fallback is never silent
            base["reference_image"] = dict(present=False, note="oracle/_ref absent: run __graft_entry__.build() next to /root/reference")
    return base, parity


def cpu_baseline_sup(wl_name, cfg, device, seconds=25.0, B=2):
    """supervised workloads: oracle model + oracle loss + backward + SGD on the host (kind "port"), and the parity of one
    HIP forward + loss in fp32 mode and in bf16 mode against it on the same weights and images"""
    from oracle import losses as o_loss
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    rng = np.random.default_rng(0)
    S = cfg.Dataset.img_size
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, 3, S, S, generator=g)
    targets = synth_targets(rng, B)
    v8 = wl_name == "v8-sup"
    if v8:
        from efficientteacher_amd.models.loss import ComputeTalLoss
        from oracle import v8 as o_v8
        ref = o_v8.Model.from_cfg(cfg)
    else:
        from efficientteacher_amd.models.loss import ComputeLoss
        from oracle import model as o_model
        ref = o_model.Model.from_cfg(cfg)
    import importlib
    mod = "efficientteacher_amd.models.detector.yolo"
    hip_items = {}
    for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        torch.manual_seed(0)
        model = importlib.import_module(mod).Model(cfg).to(device).train()
        model.set_compute_dtype(dt)
        if name == "fp32":
            miss = ref.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()}, strict=False)
            assert not miss.unexpected_keys and all(k.startswith("det_") for k in miss.missing_keys), miss   # the oracle carries the (unused) netD
        closs = (ComputeTalLoss if v8 else ComputeLoss)(model, cfg)
        if not v8:
            hp_w = (float(closs.box_w), float(closs.obj_w), float(closs.cls_w))
        loss, items = closs(model(x.to(device)), targets.to(device))
        hip_items[name] = {k: float(v) for k, v in items.items() if torch.is_tensor(v) and v.numel() == 1}
        del model, closs, loss, items
        torch.cuda.empty_cache()
    ref.train()
    opt = torch.optim.SGD(ref.parameters(), lr=0.01, momentum=0.937, nesterov=True)
    def step():
        opt.zero_grad()
        out = ref(x)
        if v8:
            rl, ri = o_v8.tal_loss(out, targets, nc=cfg.Dataset.nc, reg_max=cfg.Loss.reg_max, img_size=S, iou_type=cfg.Loss.iou_type,
                                   w_class=cfg.Loss.qfl_loss_weight, w_iou=cfg.Loss.box_loss_weight, w_dfl=cfg.Loss.dfl_loss_weight)
            ri = {k: float(ri[k].detach()) for k in ("loss_iou", "loss_dfl", "loss_cls")}
        else:
            rl, ri = o_loss.compute_loss(out[0] if isinstance(out, tuple) else out, targets, ref.head.anchors, nc=cfg.Dataset.nc,
                                         box_w=hp_w[0], obj_w=hp_w[1], cls_w=hp_w[2])
            ri = {k: float(torch.as_tensor(ri[k]).detach()) for k in ("box", "obj", "cls")}
        rl.backward()
        opt.step()
        return ri
    t0 = time.time()
    want = step()
    warm = time.time() - t0
    parity = dict(against=f"oracle ({'oracle/v8.py written spec: loss parity UNPINNED, see SURVEY.md 8 a-14' if v8 else 'oracle/model.py + oracle/losses.py'}), "
                          f"same weights and inputs, {B} images, forward + loss",
                  tolerance="fp32 mode 1e-4 (v8: 1e-3), bf16 mode 5e-2 on the loss terms")
    for name in ("fp32", "bf16"):
        rel = {k: abs(hip_items[name].get(k, float('nan')) - v) / max(abs(v), 1e-12) for k, v in want.items()}
        parity[name] = dict(loss_rel_dev={k: round(v, 7) for k, v in rel.items()}, max_loss_rel_dev=max(rel.values()))
    t0, n = time.time(), 0
    if warm < seconds:
        while n < 1 or (time.time() - t0 + warm < seconds and n < 8):
            step(); n += 1
        dt = (time.time() - t0) / n
    else:
        n, dt = 1, warm
    base = dict(value=B / dt, unit="images/s", cores=cores, kind="port",
                sample=f"{WORKLOADS[wl_name]['name']}: forward + loss + backward + SGD, batch {B} {S}x{S}, {n} steps, plain-torch fp32 CPU port (oracle/)")
    return base, parity


def _respawn(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU over RCCL), exactly
    as the documented torch.distributed.run command would."""
    import subprocess
    ndev = torch.cuda.device_count()
    if ndev < a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {ndev} GPU(s) visible -- refusing to report a {ndev}-GPU number as {a.gpus}")
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


# ---- one measured configuration ----------------------------------------------------------------------------------------
def measure(a, wl_name, per_rank, device, rank, world, dev_index, full=True):
    """warm-up + timed region + (full=True) the instrumented extras for ONE per-rank batch size; returns a dict of results.
    Collective: every rank calls this with the same arguments."""
    from efficientteacher_amd import _lib, ops
    wl = WORKLOADS[wl_name]
    ssod = wl["kind"] == "ssod"
    cfg, tr = build_trainer(device, rank, world, dev_index, per_rank, wl, a.dtype)
    rng = np.random.default_rng(1234 + rank)
    S = cfg.Dataset.img_size
    Bl = Bu = per_rank
    if ssod:
        imgs, targets, u_str, u_ori, M_s = make_batch(rng, Bl, Bu, S, device)
        if not a.float_inputs:
            # resident inputs = what the loaders deliver (uint8 NCHW): the `/ 255` of ssod_trainer.py:694 then runs INSIDE
            # the timed step (the pack kernel's IEEE division, bit-equal to torch's), as it does for the supervised workloads below
            imgs, u_str, u_ori = [(t * 255).round().to(torch.uint8) for t in (imgs, u_str, u_ori)]
        g = torch.Generator(device="cpu").manual_seed(99 + rank)
        synth = synth_teacher_scores(cfg, Bu, S, g).to(device)

        def hook(tp):           # a random-init teacher scores nothing above 0.1: SURVEY.md 8(d) synthetic scores
            tp[..., 4:] = synth
            return tp
        tr.teacher_pred_hook = hook
        tr.overlap_teacher = not a.no_overlap
        if getattr(a, "teacher_after", None) is not None:
            tr.teacher_after = "" if a.teacher_after == "start" else a.teacher_after
        if a.graph:
            tr.use_graph = True
        imgs_per_step = Bl + Bu
    else:
        g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
        imgs = torch.randint(0, 256, (per_rank, 3, S, S), generator=g, dtype=torch.uint8).to(device)   # what the loader delivers
        targets = synth_targets(rng, per_rank).to(device)
        imgs_per_step = per_rank
    ni = 2000               # past the warm-up ramp's first iterations, inside warm-up like early training

    feed = None
    if a.host_inputs and ssod:  # what a data loader hands over: uint8 NCHW batches in host memory, every step
        from efficientteacher_amd.utils.prefetch import DevicePrefetcher
        host = [(t if t.dtype == torch.uint8 else (t * 255).round().to(torch.uint8)).cpu() for t in (imgs, u_str, u_ori)]

        def batches():
            while True:
                yield host
        feed = DevicePrefetcher(batches(), device)      # copy stream, one step ahead; /255 happens in the pack kernel

    def step(i):
        if not ssod:
            return tr.train_step(imgs, targets, ni + i)
        if feed is not None:
            im, us, uo = next(feed)
            return tr.train_instance(im, targets, None, us, uo, None, M_s, ni + i)
        return tr.train_instance(imgs, targets, None, u_str, u_ori, None, M_s, ni + i)

    for i in range(a.warmup):
        step(i)
    n_extra = 0
    if ssod and getattr(tr, "use_graph", False):
        # the captured step graph is instantiated after `graph_warmup` eager steps; its first replays carry one-off costs (graph
        # upload; measured ~0.3 s in the first process on a fresh box) and the next ones are the trainer's slow-replay probe
        # (trainer/graph_step.py: a graph that replays slower than the eager step is captured once more, then dropped).  All of
        # that stays out of the timed region -- a FIXED number of steps (every rank issues the same collectives), whatever W is
        from efficientteacher_amd.trainer.graph_step import StepGraph
        need = tr.graph_warmup + 2 * (1 + StepGraph.REPLAY_PROBE) + 1
        while a.warmup + n_extra < need:
            step(a.warmup + n_extra)
            n_extra += 1
        # ... and ONE eager step after the capture: torch.cuda.graph() empties the caching allocator when it starts capturing, so the
        # instrumented eager step of the timed region would otherwise re-allocate every activation with hipMalloc inside the timed
        # region -- seen as a constant ~0.6 s (32+32) / ~0.2 s (16+16) added to some processes' timed regions (86 ms per step
        # with every replay at 54 ms by its own HIP events; NOTEBOOK.md "slow replay mode", second entry)
        if getattr(tr, "use_graph", False):
            tr.use_graph = False
            step(a.warmup + n_extra)
            n_extra += 1
            tr.use_graph = True
        torch.cuda.synchronize()
    timer = ops.KernelTimer()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # HIP-event timing of the conv launches (roofline leg) on a sample of the timed steps: two events per
    # launch are ~870 extra stream operations per step, ~3 % of the step if every step is instrumented
    timed = {a.steps // 2} if a.steps > 2 else set(range(a.steps))
    sync()
    t0 = time.perf_counter()
    ddp = tr.model if hasattr(tr.model, "collect_timing") else None
    graph_default = getattr(tr, "use_graph", False)
    items = None
    step_walls = [t0] if os.environ.get("ET_BENCH_STEP_TIMES") else None     # debugging aid: host wall clock after every enqueue
    for i in range(a.steps):
        ops.TIMER = timer if (i in timed and full) else None
        if ssod:
            tr.use_graph = graph_default and not (i in timed and full)   # HIP events cannot be recorded inside a replayed graph:
        if ddp is not None:                                              # the instrumented step(s) of the timed region run eagerly
            ddp.timing = i in timed
        items = step(a.warmup + i)
        if step_walls is not None:
            step_walls.append(time.perf_counter())
        if graph_default and getattr(tr, "graph_error", None):
            graph_default = False                                        # capture was rejected: the trainer fell back to eager steps
    ops.TIMER = None
    if ssod:
        tr.use_graph = graph_default
    t_enq = time.perf_counter() - t0          # host time to enqueue the K steps (no device sync inside a step)
    sync()
    if step_walls is not None:
        print("[step enqueue ms]", [round((b - a_) * 1e3, 1) for a_, b in zip(step_walls, step_walls[1:])],
              "drain", round((time.perf_counter() - step_walls[-1]) * 1e3, 1), file=sys.stderr)
    dt = time.perf_counter() - t0
    t_ar = ddp.collect_timing() if ddp is not None else None
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    vals = items.values() if isinstance(items, dict) else [items]
    res = dict(per_rank=per_rank, dt=dt, t_enq=t_enq, imgs_per_step=imgs_per_step, cfg=cfg,
               loss_ok=all(math.isfinite(float(v)) for v in vals if torch.is_tensor(v) and v.numel() == 1),
               loss_scale=(tr.scaler.get_scale() if getattr(tr, "scaler", None) is not None and tr.scaler.enabled else None),
               t_ar=t_ar, grad_bytes=int(tr.model.flat_state().grads.numel() * 4), graph_default=bool(graph_default),
               graph_replays=(getattr(tr._graph, "replays", 0) if getattr(tr, "_graph", None) else 0),
               graph_recaptures=(getattr(tr._graph, "recaptures", 0) if getattr(tr, "_graph", None) else 0),
               graph_probe=((getattr(tr._graph, "probe_ms", None), tr.eager_step_ms(), getattr(tr._graph, "slow_captures", 0))
                            if getattr(tr, "_graph", None) else None),
               graph_error=getattr(tr, "graph_error", None), graph_requested=bool(a.graph and ssod), graph_extra_warmup=n_extra,
               n_timed=len(timed), timer=timer, S=S)
    if not full:
        del tr
        torch.cuda.empty_cache()
        return res
    # outside the timed region: what issuing ONE step costs the host when the HIP queue is empty at its start (inside the
    # timed loop the host runs ahead until the queue is full and then advances at the GPU's pace, so t_enq ~ dt there)
    enq_empty = []
    for j in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step(a.warmup + a.steps + j)
        enq_empty.append((time.perf_counter() - t1) * 1e3)
    torch.cuda.synchronize()
    res["enq_empty"] = sorted(enq_empty)[1]
    # also outside the timed region: ONE step with every launching library call bracketed by HIP events (time budget by kernel
    # family and stream), in the overlap mode of the timed region
    fam = None
    try:
        ft = ops.FamilyTimer()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if ssod:
            tr.use_graph = False
        _lib.CALL_TIMER = ft
        e0.record()
        step(a.warmup + a.steps + 3)
        e1.record()
        _lib.CALL_TIMER = None
        torch.cuda.synchronize()
        main_id = torch.cuda.current_stream().cuda_stream
        by_stream = ft.summary()
        span = e0.elapsed_time(e1)
        main = by_stream.pop(main_id, {})
        wq_side = ops.WGRAD_QUEUE._side.get(device) if hasattr(ops.WGRAD_QUEUE, "_side") else None
        wgrad_side = by_stream.pop(wq_side.cuda_stream, {}) if wq_side is not None else {}
        side = {}
        for d in by_stream.values():
            for k, v in d.items():
                side[k] = side.get(k, 0.0) + v
        fam = dict(main_stream={k: round(v, 3) for k, v in sorted(main.items())},
                   teacher_stream={k: round(v, 3) for k, v in sorted(side.items())},
                   wgrad_stream={k: round(v, 3) for k, v in sorted(wgrad_side.items())},
                   wgrad_stream_note="the deferred, grouped weight-gradient launches run on their own HIP stream beside the dgrad / BatchNorm-"
                                     "backward chain (ops.WgradQueue, default since r06; ET_WGRAD_STREAM=0 puts them back on the main stream: "
                                     "profiles/r06_wgrad_side_stream_ab.txt); durations under contention, like the teacher stream's",
                   main_stream_span_ms=round(span, 3), main_stream_busy_ms=round(sum(main.values()), 3),
                   teacher_stream_ms=round(sum(side.values()), 3),
                   aten_and_gaps_ms=round(span - sum(main.values()), 3),
                   note="one instrumented step after the timed region; an event pair spans from the retirement of the stream's previous "
                        "command to the end of the call, so the families of a stream sum to its busy time; aten_and_gaps = span of the "
                        "main stream minus that sum (torch's own kernels: gradient-branch adds, fills, copies; launch gaps; the HIP-event "
                        "overhead of this instrumentation, ~2 x 1000 events)")
    except Exception as e:          # never lose the throughput line to an instrumentation leg
        _lib.CALL_TIMER = None
        fam = dict(error=f"{type(e).__name__}: {e}")
    finally:
        if ssod:
            tr.use_graph = graph_default
    res["families"] = fam
    # ONE instrumented step with the teacher AND the weight gradients on the main stream (no kernel shares the GPU with the timed one):
    # the same launches' durations without the inflation the co-resident streams cause
    solo = None
    if ssod:
        keep_side = ops.WGRAD_QUEUE.use_side
        try:
            t_solo = ops.KernelTimer()
            keep_overlap = tr.overlap_teacher
            tr.overlap_teacher, ops.TIMER, tr.use_graph = False, t_solo, False
            ops.WGRAD_QUEUE.use_side = 0
            step(a.warmup + a.steps + 4)
            ops.WGRAD_QUEUE.use_side = keep_side
            ops.TIMER, tr.overlap_teacher, tr.use_graph = None, keep_overlap, graph_default
            torch.cuda.synchronize()
            solo = t_solo.summary()
        except Exception:
            ops.TIMER = None
            ops.WGRAD_QUEUE.use_side = keep_side
    res["solo"] = solo
    # ... and the teacher's forward + decode ALONE on the GPU (in the step it shares the CUs with the student's forward, so the
    # teacher_stream figures above are durations under contention, not cost)
    if ssod and isinstance(fam, dict) and "error" not in fam and not a.no_teacher_alone:
        try:
            with torch.no_grad():
                for _ in range(2):
                    tr.ema.ema(u_ori, augment=False)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    tr.ema.ema(u_ori, augment=False)
                e1.record()
            torch.cuda.synchronize()
            fam["teacher_forward_alone_ms"] = round(e0.elapsed_time(e1) / 5, 3)
        except Exception as e:
            fam["teacher_forward_alone_ms"] = f"{type(e).__name__}: {e}"
    del tr
    torch.cuda.empty_cache()
    return res


def roofline_of(res, dump=None):
    timer, n_timed = res["timer"], res["n_timed"]
    agg = timer.summary()
    if dump:
        with open(dump, "w") as f:
            json.dump([dict(kernel=t, flops=fl, launches=n, bytes=nb, ms=ea.elapsed_time(eb), shape=sh)
                       for (t, fl, n, nb, ea, eb), sh in zip(timer.rows, timer.shapes)], f)
    dom = max((k for k in agg if k.startswith("conv_gemm") and "parity classes" not in k), key=lambda k: agg[k]["ms"])
    d = agg[dom]
    per_launch_flops = d["flops"] / d["launches"]
    per_launch_s = d["ms"] * 1e-3 / d["launches"]
    conv_ms = sum(v["ms"] for v in agg.values()) / n_timed
    conv_fl = sum(v["flops"] for v in agg.values()) / n_timed
    # HBM/fabric traffic of the same kernel from the committed PMC passes over this very command
    # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs; tools/pmc_summarize.py, tools/pmc_to_traffic.py)
    traffic, traffic_src = None, None
    try:
        pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if dom in pt["kernels"]:
            traffic, traffic_src = pt["kernels"][dom]["bytes_per_launch"], pt["source"]
    except (OSError, ValueError, KeyError):
        pass
    # MFMA utilisation of the same kernel as the HARDWARE counts it (north_star: "rocprof ... MFMA utilisation"): matrix-pipe busy
    # cycles over (4 SIMDs x 256 CUs x kernel cycles) from a committed `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ...` pass over this
    # very command (tools/pmc_mfma.py -> profiles/pmc_mfma.json).  `frac` above it is FLOPs / time / 2.5 PFLOP/s at the 2.4 GHz the
    # peak is quoted at; the counter fraction is against the clock the kernel actually ran at, so it reads higher by max / effective.
    mfma = None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_mfma.json")))
        km = pm["kernels"].get(dom)
        if km and km.get("mfma_busy_frac") is not None:
            mfma = dict(mfma_busy_frac=km["mfma_busy_frac"], cu_busy_frac=km.get("cu_busy_frac"), mfma_busy_of_cu_busy=km.get("mfma_busy_of_cu_busy"),
                        wait_inst_frac=km.get("wait_inst_frac"),
                        avg_us_in_that_run=km.get("avg_us"), effective_clock_ghz=pm.get("effective_clock_ghz"), clock_source=km.get("clock_source"),
                        mfma_cycles_counted_over_expected=km.get("mfma_cycles_counted_over_expected"), source=pm.get("source"),
                        top_kernels={k: round(pm["kernels"][k]["mfma_busy_frac"], 4) for k in pm.get("top_by_time", [])[:8]
                                     if pm["kernels"][k].get("mfma_busy_frac") is not None})
    except (OSError, ValueError, KeyError):
        pass
    roof = dict(bound="mfma", kernel=dom, achieved=per_launch_flops / per_launch_s / 1e12, peak=PEAK_BF16 / 1e12,
                unit="TFLOP/s", frac=per_launch_flops / per_launch_s / PEAK_BF16, traffic=traffic,
                traffic_unit="bytes per launch (2*FETCH_SIZE + WRITE_SIZE)", traffic_source=traffic_src,
                algorithmic_bytes_per_launch=d["bytes"] / d["launches"],
                launches_per_step=d["launches"] / n_timed, instrumented_steps=n_timed,
                avg_launch_us=per_launch_s * 1e6, algorithmic_gflop_per_launch=per_launch_flops / 1e9,
                all_conv_kernels=dict(ms_per_step=conv_ms, tflops=conv_fl / (conv_ms * 1e-3) / 1e12,
                                      algorithmic_tflop_per_step=conv_fl / 1e12))
    roof["mfma_busy_frac"] = mfma["mfma_busy_frac"] if mfma else None
    roof["mfma_counters"] = mfma
    solo = res.get("solo")
    if solo and dom in solo:
        sd = solo[dom]
        roof["same_kernel_nothing_co_resident"] = dict(
            avg_launch_us=sd["ms"] * 1e3 / sd["launches"], frac=sd["flops"] / (sd["ms"] * 1e-3) / PEAK_BF16,
            all_conv_ms=sum(v["ms"] for v in solo.values()),
            note="one extra step outside the timed region with the teacher forward AND the weight gradients on the main stream: in the "
                 "timed region their launches share the GPU with the student's forward / dgrad chain and lengthen them (the step is faster "
                 "with them side by side)")
        roof["same_kernel_teacher_not_overlapped"] = roof["same_kernel_nothing_co_resident"]      # the key of rounds 2-5
    return roof, conv_fl


def roofline_power(executed_flops, hbm_bytes, step_s):
    """The third roofline of this part: ENERGY.  The step runs with the package power tracker as its active limiter (profiles/
    r06_power_limit.txt: amd-smi PPT violations accumulate, no thermal ones; 1.15-1.28 kW sampled), so its floor is
    (executed FLOPs x energy per MFMA flop + HBM bytes x energy per byte) / (power cap - idle power), with the per-operation energies
    MEASURED on the part (profiles/energy_model.json <- tools/probe/kernel_power.py).  `frac` = that floor / the measured step;
    `implied_mean_power_w` = the package power the same energy needs at the measured step time (compare with rocm-smi's samples)."""
    try:
        m = json.load(open(os.path.join(ROOT, "profiles", "energy_model.json")))
        e = executed_flops * m["pj_per_mfma_flop"] * 1e-12 + (hbm_bytes or 0.0) * m["pj_per_hbm_byte"] * 1e-12
        floor = e / (m["cap_w"] - m["idle_w"])
        return dict(bound="power", energy_j_per_step=round(e, 2), mfma_j=round(executed_flops * m["pj_per_mfma_flop"] * 1e-12, 2),
                    hbm_j=round((hbm_bytes or 0.0) * m["pj_per_hbm_byte"] * 1e-12, 2), cap_w=m["cap_w"], idle_w=m["idle_w"],
                    floor_ms=round(floor * 1e3, 2), frac=round(floor / step_s, 4),
                    implied_mean_power_w=round(e / step_s + m["idle_w"], 0),
                    pj_per_mfma_flop=m["pj_per_mfma_flop"], pj_per_hbm_byte=m["pj_per_hbm_byte"], source="profiles/energy_model.json",
                    note="energy floor of the step at the package power cap; the step's clock (2.04-2.09 GHz of 2.4) is set by this limiter, "
                         "so roofline.frac against 2.5 PFLOP/s and this figure describe the same distance from two sides")
    except Exception as ex:
        return dict(bound="power", error=f"{type(ex).__name__}: {ex}"[:200])


def roofline_hbm(workload, per_rank, step_s):
    """the HBM side of the step: bytes moved per step (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over EVERY kernel of this very
    command, separate passes, FETCH doubled per the gfx950 note: profiles/pmc_traffic.json, tools/pmc_to_traffic.py) divided by
    the step time measured in THIS run, against the HBM3E peak and against the best streaming pass measured on the part; beside
    it the algorithmic bytes of SURVEY.md 8(d) and the ratio.  The traffic figure is from the committed PMC file (counters cannot
    be read inside the timed region); it is null when that file does not belong to this workload / batch."""
    out = dict(bound="hbm", peak=PEAK_HBM / 1e12, achievable=ACHIEVABLE_HBM / 1e12, unit="TB/s", achieved=None, frac=None,
               frac_of_achievable=None, traffic=None, traffic_unit="bytes per step, all kernels (2*FETCH_SIZE + WRITE_SIZE)", traffic_source=None)
    if workload == "v5l-ssod":
        alg = algorithmic_bytes_per_step(per_rank, per_rank)
        out["algorithmic_bytes_per_step"] = {k: round(v) for k, v in alg.items()}
        out["algorithmic_achieved"] = alg["total"] / step_s / 1e12
        out["algorithmic_frac"] = alg["total"] / step_s / PEAK_HBM
    try:
        pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        st = pt.get("step")
        if st and pt.get("workload", "v5l-ssod") == workload and pt.get("per_rank", 32) == per_rank:
            b = float(st["bytes_per_step"])
            out.update(traffic=b, traffic_source=pt["source"], traffic_by_family={k: round(v) for k, v in st["by_family"].items()},
                       achieved=b / step_s / 1e12, frac=b / step_s / PEAK_HBM, frac_of_achievable=b / step_s / ACHIEVABLE_HBM)
            if "algorithmic_bytes_per_step" in out:
                out["traffic_over_algorithmic"] = b / out["algorithmic_bytes_per_step"]["total"]
    except (OSError, ValueError, KeyError):
        pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="v5l-ssod", choices=sorted(WORKLOADS))
    ap.add_argument("--per-rank", type=int, default=0, help="images per rank (SSOD: labeled = unlabeled = this); default: the "
                    "BASELINE config of the workload (v5l-ssod: 32, and 16 at --gpus 8 = configs[3], global 128+128)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"], help="compute dtype of the timed step: bf16 (default performance "
                    "mode) or fp16 (the reference's autocast dtype, with the device-resident loss scaler; same MFMA rate)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-teacher-alone", action="store_true", help="skip the teacher-forward-alone timing after the timed region (profiling "
                    "runs: its seven extra forwards would count into the per-step kernel totals)")
    ap.add_argument("--no-weak-point", action="store_true", help="--gpus 8: skip the second (32+32 per rank) measurement")
    ap.add_argument("--no-overlap", action="store_true", help="run the teacher on the main stream (A/B)")
    ap.add_argument("--set", action="append", default=[], metavar="MODULE.NAME=VALUE", help="A/B: set a module-level constant of the package before the "
                    "trainer is built, e.g. --set autograd.FUSE_BN_BWD_K=7 (int / float / str literals)")
    ap.add_argument("--teacher-after", default=None, choices=["start", "p1", "p2", "p3", "p4"], help="A/B: where in the student's forward the teacher stream "
                    "starts (trainer default p3: behind the stride-8 stage)")
    ap.add_argument("--graph", action="store_true", help="replay the step as one captured HIP graph (trainer/graph_step.py) instead of "
                    "issuing every launch from Python (A/B: the 32+32 step is GPU-bound; default ON for per-rank batches < 32)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--float-inputs", action="store_true", help="A/B: resident fp32 images already divided by 255 (the form until r05) instead "
                    "of the loaders' uint8 batches")
    ap.add_argument("--host-inputs", action="store_true", help="PCIe-inclusive variant: every step receives its three "
                    "uint8 image batches from pinned host memory (never the reported `value`; noted in DESIGN.md)")
    ap.add_argument("--dump-launches", default=None, help="write (kernel, flops, bytes, ms) of every timed conv launch of the "
                    "instrumented step to this JSON file (tools/launch_table.py prints it)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); 'gloo' lets two "
                    "ranks share ONE GPU to exercise the N>1 code path where only a single GPU is available")
    ap.add_argument("--force-dp", action="store_true", help="--gpus 1 only: run the data-parallel code path (broadcasts, chunked "
                    "asynchronous AVG all-reduce from the gradient-ready hook, graph capture of the collectives) over a ONE-rank "
                    "RCCL group -- the way to execute that path on a single-GPU box")
    a = ap.parse_args()
    for kv in a.set:
        import ast
        import importlib
        name, val = kv.split("=", 1)
        parts = name.split(".")
        m = importlib.import_module("efficientteacher_amd." + parts[0])          # module(s), then attributes (ops.WGRAD_QUEUE.group)
        for i, q in enumerate(parts[1:-1], 1):
            try:
                m = importlib.import_module("efficientteacher_amd." + ".".join(parts[:i + 1]))
            except ImportError:
                m = getattr(m, q)
        assert hasattr(m, parts[-1]), f"--set: {name} does not exist"
        setattr(m, parts[-1], ast.literal_eval(val))

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _respawn(a)
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    ndev = torch.cuda.device_count()
    if a.backend == "nccl" and ndev < world:
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {ndev} GPU(s) visible (use --backend gloo to share one GPU for a functional check)")
    dev_index = local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    from efficientteacher_amd.parallel import apply_rccl_knobs
    rccl_env = apply_rccl_knobs()              # ET_RCCL_CHANNELS & co: before the communicator exists
    if world > 1 or a.force_dp:
        if a.force_dp and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29500 + (os.getpid() % 2000)))
            os.environ["ET_DP_SINGLE_RANK"] = "1"
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
        elif a.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(a.backend)

    from efficientteacher_amd import ops
    wl = WORKLOADS[a.workload]
    ssod = wl["kind"] == "ssod"
    per_rank = a.per_rank or (16 if (ssod and world == 8) else wl["per_rank"])
    if ssod and not a.no_graph and per_rank < 32 and a.backend == "nccl":
        a.graph = True          # small per-GPU batches: the ~16 ms of host enqueue are no longer small against the GPU step
    res = measure(a, a.workload, per_rank, device, rank, world, dev_index, full=True)
    weak = None
    if ssod and world == 8 and per_rank != 32 and not a.no_weak_point and not a.per_rank:
        keep_graph = a.graph
        a.graph = False
        weak = measure(a, a.workload, 32, device, rank, world, dev_index, full=False)
        a.graph = keep_graph

    if rank == 0:
        cfg, dt, S = res["cfg"], res["dt"], res["S"]
        ips = res["imgs_per_step"]
        try:                 # the roofline leg must never cost the throughput line
            roof, conv_fl = roofline_of(res, a.dump_launches)
        except Exception as e:
            roof, conv_fl = dict(bound="mfma", kernel=None, achieved=None, peak=PEAK_BF16 / 1e12, unit="TFLOP/s", frac=None,
                                 traffic=None, error=f"{type(e).__name__}: {e}"), None
        if a.workload == "v5l-ssod":
            alg_flop = F_IMG * per_rank + 3 * F_IMG * (2 * per_rank)                     # the number BASELINE quotes
            da = bool(cfg.SSOD.with_da_loss)
            step_flop = alg_flop - (0.0 if da else 2 * F_NETD_IMG * (2 * per_rank))      # EXECUTED: no netD backward without the DA loss
        else:
            alg_flop = step_flop = conv_fl if conv_fl else float("nan")      # measured: sum of 2*M*N*K over every conv launch of the step
        if ssod:
            base_cfg = "BASELINE configs[2]" if (world == 1 and per_rank == 32) else \
                       ("BASELINE configs[3]: global 128 labeled + 128 unlabeled over 8 ranks" if (world == 8 and per_rank == 16) else
                        ("configs[2]'s per-GPU batch on every rank (weak scaling)" if per_rank == 32 else "reduced per-rank batch"))
            workload = f"{wl['name']}, {per_rank} labeled + {per_rank} unlabeled 640px per GPU ({base_cfg})"
        else:
            workload = f"{wl['name']}, batch {per_rank} per GPU, 640px, synthetic COCO-80 targets"
        t_ar = res["t_ar"]
        out = {
            "metric": wl["metric"],
            "value": world * ips * a.steps / dt, "unit": "images/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if (ssod and world == 8 and per_rank == 16) else "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic (random-init weights" + ("; teacher obj/cls scores "
            "replaced by U^16 / U^4 so that NMS and the pseudo-label loss do representative work)" if ssod else "; uniform uint8 images, synthetic COCO-80 targets)"),
            "config": {"workload": workload, "global_batch": world * ips, "img_size": S,
                       "parallelism": f"dp{world}", "optimizer_every_step": True,
                       "algorithmic_tflop_per_step_per_gpu": alg_flop / 1e12,
                       "executed_tflop_per_step_per_gpu": step_flop / 1e12,
                       "flop_note": "algorithmic = SURVEY.md 8(d) (teacher F + student 3F, netD included); executed = the same minus the netD "
                                    "backward, which SSOD.with_da_loss False never runs; step_tflops / frac_of_bf16_mfma_peak use EXECUTED",
                       "step_tflops_per_gpu": step_flop / (dt / a.steps) / 1e12,
                       "frac_of_bf16_mfma_peak": step_flop / (dt / a.steps) / PEAK_BF16, "loss_finite": res["loss_ok"], "loss_scale_after_the_timed_region": res.get("loss_scale"),
                       "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1, "env_knobs": ops.env_knobs(), "module_overrides": list(a.set),
                       "rccl_env": rccl_env or None,
                       "grad_allreduce": (dict(bytes=int(sum(b for b, _ in t_ar[2])), arena_bytes=res["grad_bytes"],
                                               wire=("bf16" if sum(b for b, _ in t_ar[2]) * 2 <= res["grad_bytes"] + 1024 else "fp32"),
                                               span_ms=t_ar[0], exposed_ms=t_ar[1],
                                               per_collective=[dict(bytes=b, exposed_ms=round(ms, 4)) for b, ms in t_ar[2]],
                                               note="span: first chunk launch (during backward) -> last collective complete; exposed: "
                                                    "compute stream waiting after backward, in total and per collective in wait order "
                                                    "(conv-weight chunks from the tail of the arena first, then biases, then BN weights)")
                                          if t_ar else None),
                       "host_enqueue_ms_per_step": res["t_enq"] / a.steps * 1e3,
                       "host_enqueue_ms_empty_queue": res.get("enq_empty"),
                       "host_enqueue_note": "per_step is measured inside the timed loop, where the host blocks on the full HIP "
                                            "queue (back-pressure: it tracks the GPU step time); empty_queue is the median host time "
                                            "to issue one step after a device synchronise, i.e. the real launch cost (a graph replay "
                                            "when step_graph.enabled)",
                       "step_graph": dict(enabled=res["graph_default"], requested=res["graph_requested"], error=res["graph_error"],
                                          replays=res["graph_replays"], recaptures=res["graph_recaptures"],
                                          extra_untimed_warmup_steps=res["graph_extra_warmup"],
                                          replay_probe=(dict(zip(("replay_ms", "eager_ms", "slow_captures"), res["graph_probe"]))
                                                        if res.get("graph_probe") else None),
                                          eager_instrumented_steps=res["n_timed"]),
                       "inputs": "host uint8 (PCIe inclusive)" if a.host_inputs else
                                 ("resident in HBM (fp32, pre-divided by 255)" if a.float_inputs else
                                  "resident in HBM (uint8 NCHW as the loaders deliver; / 255 inside the timed step)")},
            "roofline": roof,
            "roofline_hbm": roofline_hbm(a.workload, per_rank, dt / a.steps),
            "kernel_ms_by_family": res.get("families"),
        }
        out["roofline_power"] = roofline_power(step_flop, (out["roofline_hbm"] or {}).get("traffic"), dt / a.steps)
        if ssod and world == 8 and per_rank == 16:
            out["scaling_note"] = ("BASELINE configs[3]: the global batch equals that of 4 ranks at configs[2]'s per-GPU batch (strong "
                                   "scaling 4 -> 8); the per-GPU-work-fixed point of the same 8 ranks is `weak_scaling_point`")
        if weak is not None:
            out["weak_scaling_point"] = dict(per_rank=f"{weak['per_rank']} labeled + {weak['per_rank']} unlabeled",
                                             value=world * weak["imgs_per_step"] * a.steps / weak["dt"], unit="images/s",
                                             ms_per_step=weak["dt"] / a.steps * 1e3, steps=a.steps, warmup=a.warmup, scaling="weak",
                                             global_batch=world * weak["imgs_per_step"])
        if world == 1 and not a.no_cpu_baseline:
            try:
                res.clear()
                torch.cuda.empty_cache()
                if ssod:
                    out["cpu_baseline"], out["parity_check"] = cpu_baseline_ssod(cfg, device)
                else:
                    out["cpu_baseline"], out["parity_check"] = cpu_baseline_sup(a.workload, cfg, device)
            except Exception as e:   # never lose the GPU number to the baseline leg
                out["cpu_baseline"] = dict(value=None, unit="images/s", cores=os.cpu_count(), kind="port",
                                           sample=f"failed: {type(e).__name__}: {e}")
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
