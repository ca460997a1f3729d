"""Parity at the BENCHMARKED configurations (BASELINE.json configs[1] and configs[2]), at their own width, depth and
resolution -- the gap VERDICT r01 led with: the bench times bf16 YOLOv5l at 640x640, the parity tests ran fp32 at
width 0.125 / 64x64.

* YOLOv5l (width 1.0, depth 1.0), 640x640, 1 labeled + 1 unlabeled image, one real SSODTrainer.train_instance on the
  HIP kernels vs oracle/step.py (the plain-torch restatement of trainer/ssod_trainer.py:587-680) on the SAME weights,
  images, M_s and injected teacher scores.
    fp32 parity mode : teacher decode 1e-3 abs (px); NMS kept indices BIT-EXACT on identical decoded inputs;
                       pseudo-label set 1e-6; student logits 2e-3 abs; the six loss terms 1e-4 relative;
                       five named gradients (from the SGD update) 5e-3 of their max.
    bf16 mode        : the same quantities at the tolerances written below (bf16 storage, fp32 accumulation).
* YOLOv5s supervised (configs[1]: width 0.5, depth 0.33), bf16, 640x640: forward + ComputeLoss + backward vs the oracle.
GPU only; the oracle legs take a few seconds of CPU each at these sizes.
"""
import copy
import os

import numpy as np
import pytest
import torch

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu
YAML = os.path.join(ROOT, "efficientteacher_amd", "configs", "ssod", "coco-standard", "yolov5l_coco_ssod_10_percent.yaml")
GRADS = ("backbone.stage1.conv.weight", "backbone.stage3_2.m.4.cv2.conv.weight", "neck.C3.m.0.cv2.conv.weight",
         "head.m.1.weight", "backbone.stage2_2.cv1.bn.weight")


@pytest.fixture
def dev():
    from efficientteacher_amd import _lib
    if not torch.cuda.is_available():
        pytest.fail("-m gpu selected but no GPU is visible")
    _lib._use_library_for_tests(None, False)
    _lib.load()
    return torch.device("cuda:0")


def _inputs(Bl, Bu, S, seed=0):
    import bench
    rng = np.random.default_rng(seed)
    imgs, targets, u_str, u_ori, M_s = bench.make_batch(rng, Bl, Bu, S, "cpu")
    g = torch.Generator().manual_seed(99)
    pw = torch.cat((torch.full((1,), 16.0), torch.full((80,), 4.0)))
    A = 3 * ((S // 8) ** 2 + (S // 16) ** 2 + (S // 32) ** 2)
    synth = torch.rand(Bu, A, 81, generator=g) ** pw
    return imgs, targets, u_str, u_ori, M_s, synth


def _canon(a):
    return a[np.lexsort((np.round(a[:, 3], 5), np.round(a[:, 2], 5), a[:, 1], a[:, 0]))]


BN_GAMMA_CONDITIONED = 0.3


def run_ssod_step_parity(dev, dtype, width=1.0, depth=1.0, S=640, Bl=1, Bu=1, with_oracle=True, amp_calibration=True, all_grads=None,
                         bn_gamma=None, deterministic=False):
    """shared by the dtype tests (and importable by tools): returns the measured deviations.
    with_oracle=False: only the HIP step (items, decoded teacher output, gradients) -- tests/test_step_benchbatch.py compares two HIP
    modes with it.  all_grads: a dict that receives {"hip": {name: grad}, "ref": {name: grad}} of EVERY parameter (the HIP ones
    recovered from the first SGD update).
    bn_gamma: every BatchNorm weight is set to this value before the step (None = the default init, 1.0).  At the default init the
    ~100-layer train-mode-BatchNorm network is CHAOTIC: rounding nothing but the conv weights to bf16, all arithmetic fp32, already
    decorrelates the weight gradients (cosine 0.2-0.3 against the unperturbed fp32 oracle; CPU bf16 autocast 0.05-0.13:
    tools/probe/grad_sensitivity.py, profiles/r04_grad_sensitivity.txt), so a gradient comparison across precisions says nothing
    about the arithmetic there.  With gamma = 0.3 the same probes give 0.992 / 0.98: the gradient bounds of the bf16 tests are
    taken at that point.
    deterministic: Model.set_deterministic(True) -- BatchNorm statistics of the 16-bit modes on the reproducible partial-row path."""
    from efficientteacher_amd.configs import get_cfg
    from efficientteacher_amd.trainer import SSODTrainer
    from efficientteacher_amd.utils.torch_utils import ModelEMA
    from oracle import model as o_model, nms as o_nms, step as o_step
    cfg = get_cfg()
    cfg.merge_from_file(YAML)
    cfg.merge_from_list(["Model.width_multiple", width, "Model.depth_multiple", depth, "Dataset.batch_size", Bl + Bu,
                         "SSOD.fixed_accumulate", True])
    cfg.freeze()
    torch.manual_seed(0)
    tr = SSODTrainer(cfg, dev, nb=1000)
    if bn_gamma is not None:
        with torch.no_grad():
            for m in tr.model.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.weight.fill_(float(bn_gamma))
    if deterministic:
        tr.model._deterministic = True           # picked up by the arena rebuild of set_compute_dtype
    tr.model.set_compute_dtype(dtype)
    assert tr.model.flat_state().deterministic == bool(deterministic)
    tr.build_optimizer(cfg)
    if dtype == torch.float16:
        # fp16 mode: the live loss scaler (optim.DeviceGradScaler).  A fixed moderate scale instead of GradScaler's initial 65536, whose
        # first steps overflow and are skipped by design: this function recovers the gradients from the FIRST update, which must
        # therefore happen (asserted below through found_inf)
        from efficientteacher_amd.optim import DeviceGradScaler
        tr.scaler = DeviceGradScaler(dev, init_scale=256.0)
    tr.ema = ModelEMA(tr.model)
    tr.semi_ema = None
    student = o_model.Model.from_cfg(cfg)
    student.load_state_dict({k: v.detach().cpu() for k, v in tr.model.state_dict().items()}, strict=True)
    teacher = copy.deepcopy(student).eval()
    student.train()
    imgs, targets, u_str, u_ori, M_s, synth = _inputs(Bl, Bu, S)
    p0 = {k: v.detach().clone().cpu() for k, v in tr.model.named_parameters()}

    captured = {}

    def hook(tp):
        tp[..., 4:] = synth.to(dev)
        captured["tp"] = tp.detach().clone().cpu()
        return tp
    tr.teacher_pred_hook = hook
    ni = 2000
    items = tr.train_instance(imgs.to(dev), targets.to(dev), None, u_str.to(dev), u_ori.to(dev), None, M_s.to(dev), ni)
    torch.cuda.synchronize()
    if dtype == torch.float16:
        assert tr.scaler.get_scale() == 256.0, "the fp16 step overflowed at scale 256 and was skipped"    # (a skip halves the scale)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    out = {}
    gp = dict(tr.model.named_parameters())
    groups = {id(p): g for g in tr.optimizer.param_groups for p in g["params"]}

    def hip_grad(name):
        # recovered from the first SGD-nesterov update: dp = -lr*(1+m)*(g + wd*p)  (buf = g on step 1)
        g = groups[id(gp[name])]
        lr, m, wd = float(g["lr"]), float(g["momentum"]), float(g["weight_decay"])
        return -(gp[name].detach().cpu() - p0[name]) / (lr * (1.0 + m)) - wd * p0[name]
    if all_grads is not None:
        all_grads["hip"] = {n: hip_grad(n) for n in gp if id(gp[n]) in groups}
    if not with_oracle:
        out["items"] = {k: float(v) for k, v in items.items()}
        out["teacher_pred"] = captured["tp"]
        return out
    ref = o_step.ssod_step(student, teacher, imgs, targets, u_str, u_ori, M_s, cfg, synth_scores=synth)
    if all_grads is not None:
        all_grads["ref"] = {n: p.grad.detach().clone() for n, p in student.named_parameters() if p.grad is not None}
    amp = None
    if dtype != torch.float32 and amp_calibration:
        # calibration: the reference's own mixed-precision recipe (autocast, trainer.py:348) on the oracle, bf16 instead
        # of fp16 -- how far reduced-precision activations move THIS step's gradients at all
        st16 = copy.deepcopy(student)
        st16.zero_grad()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            o_step.ssod_step(st16, teacher, imgs, targets, u_str, u_ori, M_s, cfg, teacher_pred=ref["teacher_pred"])
        amp = dict(st16.named_parameters())
    # 1 teacher decode (eval forward of the EMA model; the injected scores are identical by construction)
    out["teacher_box_abs"] = (captured["tp"][..., :4] - ref["teacher_pred"][..., :4]).abs().max().item()
    # 2 NMS on bit-identical inputs: the oracle re-run on the GPU's decoded tensor must keep the same rows
    from efficientteacher_amd.utils.general import nms_ssod_padded
    dets, counts, keep, _ = nms_ssod_padded(captured["tp"].to(dev), cfg.SSOD.nms_conf_thres, cfg.SSOD.nms_iou_thres)
    rdets, rkeep = o_nms.non_max_suppression_ssod(captured["tp"].numpy(), cfg.SSOD.nms_conf_thres, cfg.SSOD.nms_iou_thres)
    out["nms_keep_equal"] = all(int(counts[i]) == rkeep[i].shape[0] and np.array_equal(keep[i, :int(counts[i])].cpu().numpy(), rkeep[i])
                                and np.array_equal(dets[i, :int(counts[i])].cpu().numpy(), rdets[i]) for i in range(Bu))
    # 3 pseudo-label set of the step (GPU teacher) vs the oracle's (CPU teacher): same images/classes, boxes close
    with torch.no_grad():
        t9, valid = tr.pseudo_label_creator.create_pseudo_label_padded(captured["tp"].to(dev), M_s.to(dev), S, S)
    mine, theirs = _canon(t9[valid.bool()].cpu().numpy()), _canon(ref["t9"])
    out["n_pseudo"] = (int(mine.shape[0]), int(theirs.shape[0]))
    if mine.shape == theirs.shape:
        out["pseudo_cls_equal"] = bool(np.array_equal(mine[:, :2], theirs[:, :2]))
        out["pseudo_box_abs"] = float(np.abs(mine[:, 2:6] - theirs[:, 2:6]).max()) if mine.size else 0.0
    # 4 losses
    mine_items = {k: float(v) for k, v in items.items()}
    out["items"] = mine_items
    out["loss_rel"] = {k: abs(mine_items[k] - r) / max(abs(r), 1e-12) for k, r in
                       {**{k: ref["sup_items"][k] for k in ("box", "obj", "cls")}, **ref["un_items"]}.items()}
    out["loss_values"] = {k: (mine_items[k], r) for k, r in {**{k: ref["sup_items"][k] for k in ("box", "obj", "cls")}, **ref["un_items"]}.items()}
    # 5 gradients (five named tensors here; every conv weight at the benchmarked batch: tests/test_step_benchbatch.py)
    rp = dict(student.named_parameters())
    out["grad_rel"], out["grad_l2"], out["grad_cos"], out["amp_l2"], out["amp_cos"] = {}, {}, {}, {}, {}
    for name in GRADS:
        if name not in gp:
            continue
        grad = hip_grad(name)
        rg = rp[name].grad
        out["grad_rel"][name] = ((grad - rg).abs().max() / rg.abs().max().clamp_min(1e-12)).item()
        out["grad_l2"][name] = ((grad - rg).norm() / rg.norm().clamp_min(1e-20)).item()
        out["grad_cos"][name] = torch.nn.functional.cosine_similarity(grad.flatten(), rg.flatten(), 0).item()
        if amp is not None:
            ag = amp[name].grad.float()
            out["amp_l2"][name] = ((ag - rg).norm() / rg.norm().clamp_min(1e-20)).item()
            out["amp_cos"][name] = torch.nn.functional.cosine_similarity(ag.flatten(), rg.flatten(), 0).item()
    return out


def test_yolov5l_640_ssod_step_fp32_vs_oracle(dev):
    r = run_ssod_step_parity(dev, torch.float32)
    print("PARITY fp32", r)
    assert r["teacher_box_abs"] <= 1e-3
    assert r["nms_keep_equal"]
    assert r["n_pseudo"][0] == r["n_pseudo"][1] and r["pseudo_cls_equal"] and r["pseudo_box_abs"] <= 1e-6
    for k, v in r["loss_rel"].items():
        assert v <= 1e-4, (k, v, r["loss_values"][k])
    for k, v in r["grad_rel"].items():
        assert v <= 5e-3, (k, v)


@pytest.mark.parametrize("Bl,Bu", [(1, 1), (2, 2)], ids=["1+1", "2+2"])
def test_yolov5l_640_ssod_step_bf16_vs_oracle(dev, Bl, Bu):
    """the dtype the bench runs.  bf16 has an 8-bit mantissa (2^-9 relative rounding per stored activation): after ~100
    conv layers the logits carry ~1e-2 relative noise, the loss terms (means over 10^4..10^6 cells) far less.
    2+2 images (nothing checked the bf16 step above 1+1): the oracle leg takes ~10 s of CPU."""
    r = run_ssod_step_parity(dev, torch.bfloat16, Bl=Bl, Bu=Bu, amp_calibration=False)
    print("PARITY bf16", {k: v for k, v in r.items() if k != "items"})
    assert r["teacher_box_abs"] <= 8.0                       # pixels, boxes up to 640 px wide
    assert r["nms_keep_equal"]                               # NMS itself is fp32 on whatever the teacher produced
    for k, v in r["loss_rel"].items():
        assert v <= 5e-2, (k, v, r["loss_values"][k])
    # gradients: test_yolov5l_640_ssod_step_bf16_gradients_vs_oracle below (at the default init they are chaotic, see bn_gamma)


def test_yolov5l_640_ssod_step_bf16_gradients_vs_oracle(dev):
    """bf16-mode weight gradients against the fp32 ORACLE, directly (the bound used to lean on a second noisy
    path, the oracle under CPU autocast).  2 + 2 images at the well-conditioned point bn_gamma = 0.3 (run_ssod_step_parity explains
    why not at the default init): EVERY conv weight's gradient has cosine >= 0.95 and relative L2 <= 0.35 against the oracle's
    (the oracle's own weights-rounded-to-bf16 probe sits at 0.992, CPU bf16 autocast at 0.98: tools/probe/grad_sensitivity.py),
    the loss terms stay within 5e-2."""
    grads = {}
    r = run_ssod_step_parity(dev, torch.bfloat16, Bl=2, Bu=2, amp_calibration=False, all_grads=grads, bn_gamma=BN_GAMMA_CONDITIONED)
    for k, v in r["loss_rel"].items():
        assert v <= 5e-2, (k, v, r["loss_values"][k])
    cos, l2 = {}, {}
    for name, rg in grads["ref"].items():
        g = grads["hip"].get(name)
        if g is None or rg.dim() != 4 or float(rg.norm()) == 0.0:
            continue
        cos[name] = torch.nn.functional.cosine_similarity(g.flatten().double(), rg.flatten().double(), 0).item()
        l2[name] = ((g - rg).norm() / rg.norm()).item()
    worst = min(cos, key=cos.get)
    vals = sorted(cos.values())
    print("PARITY bf16 gradients vs fp32 oracle (bn_gamma 0.3, 2+2):", len(vals), "conv tensors; cosine min", (worst, cos[worst]), "median",
          vals[len(vals) // 2], "; worst relative L2", max(l2.values()), "; loss_rel", r["loss_rel"])
    assert len(vals) >= 100
    assert vals[0] >= 0.95, (worst, cos[worst])
    assert max(l2.values()) <= 0.35, max(l2, key=l2.get)


def test_yolov5l_640_ssod_step_fp16_deterministic_vs_oracle_and_twice(dev):
    """with the reproducibility switch (Model.set_deterministic / cfg.Model.deterministic_bn) the fp16 step
    holds the r04 bound -- every loss term within 5e-3 of the fp32 oracle (measured 3.5e-3, the same value every run) -- and two runs of
    the step from the same state give BIT-EQUAL loss items and a bit-equal decoded teacher output; the bf16 step likewise twice."""
    r = run_ssod_step_parity(dev, torch.float16, Bl=2, Bu=2, amp_calibration=False, deterministic=True)
    print("PARITY fp16 deterministic 2+2:", {k: r[k] for k in ("teacher_box_abs", "nms_keep_equal", "n_pseudo", "loss_rel")})
    assert r["teacher_box_abs"] <= 5e-2 and r["nms_keep_equal"]
    for k, v in r["loss_rel"].items():
        assert v <= 5e-3, (k, v, r["loss_values"][k])
    # twice from the same state: the loss ITEMS are sums over workgroups collected with fp32 atomics (loss.hip), so they agree to
    # the last bits of that addition, not bit for bit; what the switch makes bit-reproducible is everything the network computes
    for dt in (torch.float16, torch.bfloat16):
        a = run_ssod_step_parity(dev, dt, Bl=2, Bu=2, with_oracle=False, deterministic=True)
        b = run_ssod_step_parity(dev, dt, Bl=2, Bu=2, with_oracle=False, deterministic=True)
        for k in a["items"]:
            assert abs(a["items"][k] - b["items"][k]) <= 2e-6 * max(abs(a["items"][k]), 1e-6), (dt, k, a["items"][k], b["items"][k])
        assert torch.equal(a["teacher_pred"], b["teacher_pred"])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_yolov5l_640_train_forward_bit_equal_in_deterministic_mode(dev, dtype):
    """the two-runs-bit-equal assertion: YOLOv5l, 2 x 640 x 640, train-mode forward twice on the reproducible
    BatchNorm path -> the three head outputs are torch.equal; in the default (sharded fp32 atomics) mode the same comparison is
    printed, not asserted (the sums differ in their last bits from run to run and ~1 stored activation in 10^4 flips)."""
    from efficientteacher_amd.configs import get_cfg
    from efficientteacher_amd.models.detector.yolo_ssod import Model
    cfg = get_cfg()
    cfg.merge_from_file(YAML)
    cfg.freeze()
    torch.manual_seed(0)
    model = Model(cfg).to(dev)
    x = torch.rand(2, 3, 640, 640, generator=torch.Generator().manual_seed(5)).to(dev)
    for det in (True, False):
        model._deterministic = det
        model.set_compute_dtype(dtype)
        model.train()
        outs = []
        for _ in range(2):
            with torch.no_grad():
                pred, _ = model(x)
            outs.append([p.detach().clone() for p in pred])
        same = all(torch.equal(a, b) for a, b in zip(*outs))
        worst = max((a.float() - b.float()).abs().max().item() for a, b in zip(*outs))
        print(f"train forward twice, {dtype}, deterministic={det}: bit-equal {same}, max |diff| {worst:.3e}")
        if det:
            assert same


def test_yolov5l_640_ssod_step_fp16_vs_oracle(dev):
    """fp16 compute mode = the REFERENCE's reduced-precision recipe (autocast to float16 + GradScaler, trainer.py:248,348,399-401) on the
    HIP kernels (v_mfma_f32_32x32x16_f16), 2 + 2 images, against the fp32 oracle: teacher decode 0.05 px, NMS keep indices bit-exact on
    identical decoded inputs, the same pseudo-label set, loss terms within 1e-2 (bf16 mode: 5e-2); the step is NOT skipped at scale 256.
    The loss bound: with the BatchNorm statistics on partial rows + fp64 finalize the step is reproducible and the worst term sits at
    3.5e-3; on the sharded fp32 accumulators (the default of the 16-bit modes since r05) the sums depend on atomic order in their last
    bits and the same term measures 0.8e-3 ... 6.0e-3 from run to run (tools/probe/fp16_loss_dev_sharded_ab.py,
    profiles/r05_fp16_loss_dev_rows_vs_sharded.txt) -- the format's own rounding noise through this network, sampled anew each run."""
    r = run_ssod_step_parity(dev, torch.float16, Bl=2, Bu=2, amp_calibration=False)
    print("PARITY fp16 2+2:", {k: r[k] for k in ("teacher_box_abs", "nms_keep_equal", "n_pseudo", "loss_rel", "grad_cos", "grad_l2")})
    assert r["teacher_box_abs"] <= 5e-2
    assert r["nms_keep_equal"]
    for k, v in r["loss_rel"].items():
        assert v <= 1e-2, (k, v, r["loss_values"][k])


def test_yolov5l_640_ssod_step_fp16_gradients_vs_oracle(dev):
    """fp16-mode weight gradients against the fp32 ORACLE at the well-conditioned point (BatchNorm weights 0.3), 2 + 2 images: EVERY conv
    weight's gradient has cosine >= 0.995 and relative L2 <= 0.1 (the bf16 mode's bounds at the same point: 0.95 / 0.35) -- with the
    reference's own arithmetic the gradient comparison VERDICT r04 (weak 1) asked for is an order of magnitude tighter."""
    grads = {}
    r = run_ssod_step_parity(dev, torch.float16, Bl=2, Bu=2, amp_calibration=False, all_grads=grads, bn_gamma=BN_GAMMA_CONDITIONED)
    for k, v in r["loss_rel"].items():
        assert v <= 1e-2, (k, v, r["loss_values"][k])      # (see test_yolov5l_640_ssod_step_fp16_vs_oracle)
    cos, l2 = {}, {}
    for name, rg in grads["ref"].items():
        g = grads["hip"].get(name)
        if g is None or rg.dim() != 4 or float(rg.norm()) == 0.0:
            continue
        cos[name] = torch.nn.functional.cosine_similarity(g.flatten().double(), rg.flatten().double(), 0).item()
        l2[name] = ((g - rg).norm() / rg.norm()).item()
    worst = min(cos, key=cos.get)
    vals = sorted(cos.values())
    print("PARITY fp16 gradients vs fp32 oracle (bn_gamma 0.3, 2+2):", len(vals), "conv tensors; cosine min", (worst, cos[worst]), "median",
          vals[len(vals) // 2], "; worst relative L2", max(l2.values()), "; loss_rel", r["loss_rel"])
    assert len(vals) >= 100
    assert vals[0] >= 0.995, (worst, cos[worst])
    assert max(l2.values()) <= 0.1, max(l2, key=l2.get)


def test_yolov5s_640_supervised_bf16_vs_oracle(dev):
    """BASELINE configs[1]: YOLOv5s supervised, bf16, 640x640 (batch 8 here: the oracle leg is CPU fp32; the bench-size
    batch 64 changes only M, which tests/test_conv.py covers per kernel instantiation)."""
    from efficientteacher_amd.configs import get_cfg
    from efficientteacher_amd.models.detector.yolo_ssod import Model
    from efficientteacher_amd.models.loss import ComputeLoss
    from oracle import losses as o_loss, model as o_model
    import bench
    cfg = get_cfg()
    cfg.merge_from_file(YAML)
    cfg.merge_from_list(["Model.width_multiple", 0.50, "Model.depth_multiple", 0.33])
    torch.manual_seed(0)
    model = Model(cfg)
    ref = o_model.Model.from_cfg(cfg)
    ref.load_state_dict(model.state_dict(), strict=True)
    model = model.to(dev).train()
    model.set_compute_dtype(torch.bfloat16)
    B = int(os.environ.get("ET_TEST_V5S_BATCH", "8"))
    rng = np.random.default_rng(3)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, 3, 640, 640, generator=g)
    targets = bench.synth_targets(rng, B)
    closs = ComputeLoss(model, cfg)
    pred, _ = model(x.to(dev))
    loss, items = closs(pred, targets.to(dev))
    loss.backward()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    rp, _ = ref.train()(x)
    rl, ritems = o_loss.compute_loss(rp, targets, ref.head.anchors, nc=80, box_w=closs.box_w, obj_w=closs.obj_w, cls_w=closs.cls_w)
    rl.backward()
    rel = {k: abs(items[k].item() - float(ritems[k])) / abs(float(ritems[k])) for k in ("box", "obj", "cls")}
    # calibration: the oracle under the reference's AMP recipe (autocast), see run_ssod_step_parity
    ref16 = copy.deepcopy(ref)
    ref16.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        rp16, _ = ref16.train()(x)
        rl16, _ = o_loss.compute_loss(rp16, targets, ref16.head.anchors, nc=80, box_w=closs.box_w, obj_w=closs.obj_w, cls_w=closs.cls_w)
    rl16.backward()
    gp, gr, ga = dict(model.named_parameters()), dict(ref.named_parameters()), dict(ref16.named_parameters())
    met = {}
    for name in ("backbone.stage1.conv.weight", "backbone.stage3_2.cv3.conv.weight", "neck.C3.m.0.cv2.conv.weight", "head.m.1.weight"):
        a, b, c = gp[name].grad.float().cpu(), gr[name].grad, ga[name].grad.float()
        met[name] = dict(l2=((a - b).norm() / b.norm()).item(), cos=torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), 0).item(),
                         amp_l2=((c - b).norm() / b.norm()).item())
    print("PARITY v5s bf16", rel, met)
    for k, v in rel.items():
        assert v <= 3e-2, (k, v)
    for name, m in met.items():
        assert m["l2"] <= 2.0 * m["amp_l2"] + 0.05, (name, m)


V8_YAML = os.path.join(ROOT, "efficientteacher_amd", "configs", "sup", "public", "yolov8m_coco.yaml")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_yolov8_full_width_640_vs_oracle(dev, dtype):
    """BASELINE configs[4]'s model at its own size: YOLOv8 (C2f backbone / PAN neck / decoupled DFL head)
    at width = depth = 1.0, 640x640, B = 2, against oracle/v8.py on the same weights -- train-mode logits, the TAL loss terms
    (written spec: loss parity unpinned), named gradients, and the eval-mode decode.  The kernel
    instantiations this model launches are pinned by name in tests/test_conv.py::test_bench_workloads_launch_only_covered_instantiations."""
    from efficientteacher_amd.configs import get_cfg
    from efficientteacher_amd.models.detector.yolo import Model
    from efficientteacher_amd.models.loss import ComputeTalLoss
    from oracle import v8 as o_v8
    import bench
    cfg = get_cfg()
    cfg.merge_from_file(V8_YAML)
    cfg.merge_from_list(["Model.width_multiple", 1.0, "Model.depth_multiple", 1.0, "Dataset.batch_size", 2])
    cfg.freeze()
    torch.manual_seed(0)
    model = Model(cfg)
    ref = o_v8.Model.from_cfg(cfg)
    ref.load_state_dict(model.state_dict(), strict=True)
    model = model.to(dev).train()
    model.set_compute_dtype(dtype)
    B, S = 2, 640
    g = torch.Generator().manual_seed(11)
    x = torch.rand(B, 3, S, S, generator=g)
    targets = bench.synth_targets(np.random.default_rng(7), B)
    closs = ComputeTalLoss(model, cfg)
    feats, cls, reg = model(x.to(dev))
    loss, items = closs((feats, cls, reg), targets.to(dev))
    loss.backward()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref.train()
    rout = ref(x)
    rl, ritems = o_v8.tal_loss(rout, targets, nc=80, reg_max=cfg.Loss.reg_max, img_size=S, iou_type=cfg.Loss.iou_type,
                               w_class=cfg.Loss.qfl_loss_weight, w_iou=cfg.Loss.box_loss_weight, w_dfl=cfg.Loss.dfl_loss_weight)
    rl.backward()
    _, rcls, rreg = rout
    fp32 = dtype == torch.float32
    d_cls = (cls.float().cpu() - rcls).abs().max().item()
    d_reg = (reg.float().cpu() - rreg).abs().max().item()
    rel = {k: abs(float(items[k]) - float(ritems[k].detach())) / max(abs(float(ritems[k].detach())), 1e-12) for k in ("loss_iou", "loss_dfl", "loss_cls")}
    gp, gr = dict(model.named_parameters()), dict(ref.named_parameters())
    names = ("backbone.stage1.conv.weight", "backbone.stage3_2.m.1.cv2.conv.weight", "neck.C3.cv2.conv.weight", "head.cv3.1.2.weight",
             "head.cv2.0.1.conv.weight")
    gl2 = {n: ((gp[n].grad.float().cpu() - gr[n].grad).norm() / gr[n].grad.norm().clamp_min(1e-20)).item() for n in names}
    l2_cls = ((cls.float().cpu() - rcls).norm() / rcls.norm()).item()
    l2_reg = ((reg.float().cpu() - rreg).norm() / rreg.norm()).item()
    print("PARITY v8", "fp32" if fp32 else "bf16", dict(d_cls=d_cls, d_reg=d_reg, l2_cls=l2_cls, l2_reg=l2_reg, loss_rel=rel, grad_l2=gl2,
                                                        num_fg=float(ritems["num_fg"])))
    assert float(ritems["num_fg"]) > 0
    if fp32:
        assert d_cls <= 2e-3 and d_reg <= 2e-3, (d_cls, d_reg)
        for k, v in rel.items():
            assert v <= 1e-3, (k, v)
        for n, v in gl2.items():
            assert v <= 2e-2, (n, v)
    else:
        # bf16 storage: logits carry ~1e-2 relative noise after ~100 conv layers (same bound as the YOLOv5l step above)
        # calibration, as for the YOLOv5l step: the oracle under the reference's own AMP recipe (autocast, bf16 for fp16).  With 2
        # images and train-mode BatchNorm at random init, bf16 rounding is amplified layer after layer (a 20x20 level normalises
        # over 800 samples); single logits move by ~2 where the values are ~14, and the DFL branch -- whose logits at init are
        # the bias 1.0 plus a small signal -- by ~10 % in relative L2 under EITHER bf16 path.
        for k, v in rel.items():
            assert v <= 5e-2, (k, v)
        ref16 = copy.deepcopy(ref)
        ref16.zero_grad()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            rout16 = ref16(x)
            rl16, _ = o_v8.tal_loss(rout16, targets, nc=80, reg_max=cfg.Loss.reg_max, img_size=S, iou_type=cfg.Loss.iou_type,
                                    w_class=cfg.Loss.qfl_loss_weight, w_iou=cfg.Loss.box_loss_weight, w_dfl=cfg.Loss.dfl_loss_weight)
        rl16.backward()
        amp_cls = ((rout16[1].float() - rcls).norm() / rcls.norm()).item()
        amp_reg = ((rout16[2].float() - rreg).norm() / rreg.norm()).item()
        print("PARITY v8 bf16 calibration (oracle under autocast)", dict(l2_cls=amp_cls, l2_reg=amp_reg))
        assert l2_cls <= 2.0 * amp_cls + 0.02 and l2_reg <= 2.0 * amp_reg + 0.02, (l2_cls, amp_cls, l2_reg, amp_reg)
        ga = dict(ref16.named_parameters())
        for n, v in gl2.items():
            amp = ((ga[n].grad.float() - gr[n].grad).norm() / gr[n].grad.norm().clamp_min(1e-20)).item()
            assert v <= 2.0 * amp + 0.05, (n, v, amp)
    # eval-mode decode (DFL expectation + dist2bbox + stride, et_v8_decode) on the same weights
    model.eval(); ref.eval()
    with torch.no_grad():
        z, _ = model(x.to(dev))
        rz, _ = ref(x)
    dz_box = (z[..., :4].cpu() - rz[..., :4]).abs().max().item()
    dz_cls = (z[..., 5:].cpu() - rz[..., 5:]).abs().max().item()
    assert dz_box <= (1e-2 if fp32 else 8.0) and dz_cls <= (1e-4 if fp32 else 2e-2), (dz_box, dz_cls)
