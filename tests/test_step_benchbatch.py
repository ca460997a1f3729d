"""Whole-step parity AT THE BENCHMARKED BATCH (BASELINE.json configs[2]: YOLOv5l, 640x640, 32 labeled + 32 unlabeled images).

tests/test_step_fullsize.py checks the step at 1 + 1 and 2 + 2 images; the kernel instantiations the library picks depend on
the batch (tile selection reads N, BatchNorm statistics are over the batch, the stride-2 dgrads split into parity classes), and
per-kernel coverage at the bench shapes (tests/test_conv.py::SELECT) is not a step check.  Here:

  (a) fp32 parity mode, 32 + 32, ONE real SSODTrainer.train_instance vs oracle/step.py on the same weights / images / M_s /
      injected teacher scores: the six loss terms <= 1e-4 relative, NMS rows and kept indices bit-exact on the identical decoded
      tensor, pseudo-label set <= 1e-6, and EVERY conv / BN / bias gradient against the oracle's (cosine >= 0.9999, relative
      L2 <= 2e-2).
  (b) bf16 performance mode (the dtype the bench line states) on the same inputs against (a)'s fp32-mode HIP step AND the fp32
      oracle: loss terms <= 5e-2.
  (c) bf16-mode gradients of all conv weights against the fp32-mode step at 32 + 32, per-tensor cosine >= 0.95 -- at a
      well-conditioned point (BatchNorm weights 0.3): at the default init the network is chaotic and no reduced-precision path,
      the reference's own autocast recipe included, keeps its gradients correlated with fp32 (tests/test_step_fullsize.py
      run_ssod_step_parity, tools/probe/grad_sensitivity.py).  No second noisy path is involved in any bound.

One oracle step at 64 images is ~25-60 s of CPU and ~100 GB of host memory (fp32 activations of 64 images kept for backward);
when the box has less than ET_TEST_BENCHBATCH_MIN_GB (default 220) of available memory the batch drops to 16 + 16 -- the
printed line says which.  GPU only.
"""
import os

import pytest
import torch

from tests.test_step_fullsize import dev, run_ssod_step_parity  # noqa: F401  (dev is a fixture)

pytestmark = pytest.mark.gpu


def _mem_available_gb():
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) / (1 << 20)
    except OSError:
        pass
    return 0.0


def _batch():
    forced = os.environ.get("ET_TEST_BENCHBATCH")
    if forced:
        return int(forced), int(forced)
    need = float(os.environ.get("ET_TEST_BENCHBATCH_MIN_GB", "220"))
    return (32, 32) if _mem_available_gb() >= need else (16, 16)


_CACHE = {}


def _fp32(dev):
    if "fp32" not in _CACHE:
        Bl, Bu = _batch()
        grads = {}
        r = run_ssod_step_parity(dev, torch.float32, Bl=Bl, Bu=Bu, all_grads=grads)
        _CACHE["fp32"] = (r, grads, (Bl, Bu))
        torch.cuda.empty_cache()
    return _CACHE["fp32"]


def _is_conv_weight(name, g):
    return g.dim() == 4


def test_fp32_step_at_the_benchmarked_batch_vs_oracle(dev):
    r, grads, (Bl, Bu) = _fp32(dev)
    print(f"PARITY bench-batch fp32 {Bl}+{Bu} (host MemAvailable {_mem_available_gb():.0f} GB)",
          {k: r[k] for k in ("teacher_box_abs", "nms_keep_equal", "n_pseudo", "pseudo_box_abs", "loss_rel")})
    assert r["teacher_box_abs"] <= 1e-3
    assert r["nms_keep_equal"]
    assert r["n_pseudo"][0] == r["n_pseudo"][1] and r["pseudo_cls_equal"] and r["pseudo_box_abs"] <= 1e-6
    for k, v in r["loss_rel"].items():
        assert v <= 1e-4, (k, v, r["loss_values"][k])
    worst_cos, worst_l2 = (None, 2.0), (None, 0.0)
    n = 0
    for name, rg in grads["ref"].items():
        g = grads["hip"].get(name)
        if g is None or float(rg.norm()) == 0.0:
            continue
        n += 1
        cos = torch.nn.functional.cosine_similarity(g.flatten().double(), rg.flatten().double(), 0).item()
        l2 = ((g - rg).norm() / rg.norm()).item()
        if cos < worst_cos[1]:
            worst_cos = (name, cos)
        if l2 > worst_l2[1]:
            worst_l2 = (name, l2)
    print("PARITY bench-batch fp32 gradients:", n, "tensors; worst cosine", worst_cos, "worst relative L2", worst_l2)
    assert n >= 300                                           # 110 conv weights + 2 x 101 BN vectors + biases
    assert worst_cos[1] >= 0.9999, worst_cos
    # measured 3.6e-3 (a BatchNorm bias of the last backbone stage).  This bound sits close to what fp32 ROUNDING alone does to this
    # net at random init: contracting ONE multiply-add in the Bottleneck shortcut (activations move by an ulp) took it to 1.16e-2
    # with the cosine at 0.99993 (r04, profiles/r04_fp32_gradient_bound_sensitivity.txt) -- so 2e-2 here, the cosine above is the
    # sharper statement
    assert worst_l2[1] <= 2e-2, worst_l2


def test_bf16_step_at_the_benchmarked_batch_vs_fp32(dev):
    """the timed dtype, default init (the weights the bench runs with): the six loss terms within 5e-2 of BOTH the fp32-mode HIP
    step and the fp32 oracle (measured ~1e-3: profiles/r04_benchbatch_parity.txt).  Gradients are compared in the next test -- at
    this init they are chaotic in ANY reduced precision (run_ssod_step_parity's bn_gamma note; the first run of this test measured
    a median cosine of 0.11 for the HIP bf16 path, the oracle under CPU autocast sits at 0.13, the oracle with bf16-rounded
    weights and fp32 arithmetic at 0.32)."""
    r32, g32, (Bl, Bu) = _fp32(dev)
    r16 = run_ssod_step_parity(dev, torch.bfloat16, Bl=Bl, Bu=Bu, with_oracle=False)
    rel_hip = {k: abs(r16["items"][k] - r32["items"][k]) / max(abs(r32["items"][k]), 1e-12) for k in r32["loss_rel"]}
    rel_ref = {k: abs(r16["items"][k] - r32["loss_values"][k][1]) / max(abs(r32["loss_values"][k][1]), 1e-12) for k in r32["loss_rel"]}
    print(f"PARITY bench-batch bf16 {Bl}+{Bu}: loss vs fp32 HIP", rel_hip, "vs oracle", rel_ref)
    for k in rel_hip:
        assert rel_hip[k] <= 5e-2 and rel_ref[k] <= 5e-2, (k, rel_hip[k], rel_ref[k])


def test_bf16_gradients_at_the_benchmarked_batch_vs_fp32(dev):
    """bf16-mode gradients of EVERY conv weight against the fp32-mode HIP step (itself pinned on the oracle above, cosine 0.99999)
    on the same 32 + 32 inputs, at the well-conditioned point bn_gamma = 0.3: cosine >= 0.95 per tensor, >= 0.975 for the median
    tensor, relative L2 <= 0.35; loss terms within 5e-2.  (bf16 storage: 2^-9 relative rounding per stored activation, ~100 layers.)"""
    from tests.test_step_fullsize import BN_GAMMA_CONDITIONED
    Bl, Bu = _batch()
    g32, g16 = {}, {}
    r32 = run_ssod_step_parity(dev, torch.float32, Bl=Bl, Bu=Bu, with_oracle=False, all_grads=g32, bn_gamma=BN_GAMMA_CONDITIONED)
    torch.cuda.empty_cache()
    r16 = run_ssod_step_parity(dev, torch.bfloat16, Bl=Bl, Bu=Bu, with_oracle=False, all_grads=g16, bn_gamma=BN_GAMMA_CONDITIONED)
    rel = {k: abs(r16["items"][k] - v) / max(abs(v), 1e-12) for k, v in r32["items"].items() if k in ("box", "obj", "cls", "ss_box", "ss_obj", "ss_cls")}
    cos, l2 = {}, {}
    for name, rg in g32["hip"].items():
        g = g16["hip"].get(name)
        if g is None or rg.dim() != 4 or float(rg.norm()) == 0.0:
            continue
        cos[name] = torch.nn.functional.cosine_similarity(g.flatten().double(), rg.flatten().double(), 0).item()
        l2[name] = ((g - rg).norm() / rg.norm()).item()
    vals = sorted(cos.values())
    worst = min(cos, key=cos.get)
    print(f"PARITY bench-batch bf16 gradients {Bl}+{Bu} (bn_gamma {BN_GAMMA_CONDITIONED}) vs the fp32-mode step:", len(vals), "conv tensors; cosine min",
          (worst, cos[worst]), "p10", vals[len(vals) // 10], "median", vals[len(vals) // 2], "; worst relative L2", max(l2.values()), "; loss", rel)
    for k, v in rel.items():
        assert v <= 5e-2, (k, v)
    assert len(vals) >= 100
    assert vals[0] >= 0.95, (worst, cos[worst])
    assert vals[len(vals) // 2] >= 0.975, vals[len(vals) // 2]
    assert max(l2.values()) <= 0.35, max(l2, key=l2.get)
