"""CPU tier: the oracle restatements (oracle/*.py) re-checked against the committed golden vectors that
oracle/make_golden.py produced by running the reference itself.  No reference tree, no GPU needed."""
import numpy as np
import torch

from oracle import assigner as o_asg
from oracle import losses as o_loss
from oracle import model as o_model
from oracle import nms as o_nms
from oracle import pseudo_label as o_pl
from tests.conftest import golden


def test_nms_oracle_vs_golden():
    g = golden("nms")
    for case in ("a", "b", "c", "empty", "tie06"):
        ct, it = g[f"{case}_thr"]
        dets, keeps = o_nms.non_max_suppression_ssod(g[f"{case}_pred"], ct, it)
        assert np.array_equal(np.concatenate(dets, 0), g[f"{case}_dets"])
        assert np.array_equal(np.concatenate(keeps, 0), g[f"{case}_keep"])
        v = o_nms.non_max_suppression(g[f"{case}_pred"], ct, it, multi_label=True)
        assert np.array_equal(np.concatenate(v, 0), g[f"{case}_val_dets"])


def test_nms_threshold_compare_rule_is_the_cuda_kernels():
    """tie06 (ORACLE-DERIVED golden, not a reference output: torchvision is absent here, oracle/nms.py header): IoU == fp32(0.6) exactly.
    The chosen rule = our reading of torchvision's CUDA kernel (fp32 threshold: the tie survives); the CPU kernel's double compare would
    suppress it -- the golden holds the chosen rule and the two rules must differ on this case.  Verification against a real torchvision:
    tests/test_nms_torchvision.py (SKIPPED in both tiers of this environment)."""
    g = golden("nms")
    bx = o_nms.xywh2xyxy(g["tie06_pred"][0, :, :4])
    sc = g["tie06_pred"][0, :, 4]
    cuda = o_nms.nms(bx, sc, 0.6)
    cpu = o_nms.nms(bx, sc, 0.6, thr_compare="cpu_double")
    assert np.array_equal(cuda, g["tie06_keep"])
    assert 1 in cuda.tolist() and 1 not in cpu.tolist() and len(cpu) == len(cuda) - 1
    # 0.65 (SSOD.nms_iou_thres): fp32(0.65) < 0.65, the rules cannot differ
    for thr in (0.65, 0.45):
        assert np.array_equal(o_nms.nms(bx, sc, thr), o_nms.nms(bx, sc, thr, thr_compare="cpu_double"))


def test_nms_ssod_options_oracle_vs_golden():
    """the optional arguments of non_max_suppression_ssod (classes, multi_label, labels, agnostic) against the reference's output"""
    g = golden("nms_ssod_options")
    rows, cnt = g["apriori_rows"], g["apriori_counts"]
    offs = np.concatenate(([0], np.cumsum(cnt)))
    labels = [rows[offs[i]:offs[i + 1]] for i in range(len(cnt))]
    variants = {"classes": dict(classes=[1, 4]), "classes_agnostic": dict(classes=[0, 2, 5], agnostic=True),
                "multi_label": dict(multi_label=True), "multi_label_classes": dict(multi_label=True, classes=[3]),
                "labels": dict(labels=labels), "labels_multi": dict(labels=labels, multi_label=True, agnostic=True)}
    for k, kw in variants.items():
        dets, _ = o_nms.non_max_suppression_ssod(g["pred"], g["thr"][0], g["thr"][1], **kw)
        assert [d.shape[0] for d in dets] == list(g[f"{k}_counts"]), k
        assert np.array_equal(np.concatenate(dets, 0), g[f"{k}_dets"]), k


def test_assigner_oracle_vs_golden():
    g = golden("assigner")
    shapes = [tuple(s) for s in g["shapes"]]
    res = o_asg.build_targets(shapes, g["anchors"], g["targets"], float(g["anchor_t"]))
    for i, r in enumerate(res):
        for k, v in r.items():
            assert np.array_equal(v, g[f"bt{i}_{k}"]), (i, k)
    resu = o_asg.build_targets(shapes, g["anchors"], g["targets7"], float(g["anchor_t"]), with_score=True)
    for i, r in enumerate(resu):
        assert np.array_equal(r["tscore"], g[f"uc{i}_tscore"]) and np.array_equal(r["gi"], g[f"uc{i}_gi"])
    # empty input (yolo_anchor_assigner.py:356-358)
    e = o_asg.build_targets(shapes, g["anchors"], np.zeros((0, 6), np.float32))
    assert all(r["b"].shape == (0,) and r["tbox"].shape == (0, 4) for r in e)


def test_ciou_and_losses_oracle_vs_golden():
    g = golden("ciou")
    pb = torch.from_numpy(g["pbox"]).requires_grad_(True)
    iou = o_loss.ciou_xywh(pb, torch.from_numpy(g["tbox"]))
    (1 - iou).mean().backward()
    assert np.array_equal(iou.detach().numpy(), g["iou"]) and np.array_equal(pb.grad.numpy(), g["grad"])
    g = golden("compute_loss")
    w = g["weights"]
    p = [torch.from_numpy(g[f"p{i}"]).requires_grad_(True) for i in range(3)]
    loss, items = o_loss.compute_loss(p, torch.from_numpy(g["targets"]), torch.from_numpy(g["anchors"]), nc=80,
                                      box_w=w[0], obj_w=w[1], cls_w=w[2], anchor_t=w[3])
    loss.backward()
    assert np.allclose(loss.detach().numpy(), g["loss"], rtol=1e-6)
    for i in range(3):
        assert np.allclose(p[i].grad.numpy(), g[f"grad{i}"], rtol=1e-5, atol=1e-8)
    gs = golden("student_match_loss")
    w = gs["weights"]
    for tag, kw in (("default", {}), ("cls", dict(with_cls=True)), ("ignore", dict(ignore_obj=True))):
        p = [torch.from_numpy(g[f"p{i}"]).requires_grad_(True) for i in range(3)]
        loss, _ = o_loss.compute_student_match_loss(p, torch.from_numpy(gs["targets9"]), torch.from_numpy(gs["anchors"]),
                                                    nc=80, box_w=w[0], obj_w=w[1], cls_w=w[2], anchor_t=w[3], **kw)
        assert np.allclose(loss.detach().numpy(), gs[f"{tag}_loss"], rtol=1e-6), tag
    gd = golden("domain_loss")
    f = [torch.from_numpy(gd[f"f{i}"]) for i in range(3)]
    assert np.allclose(o_loss.domain_loss(f, 0).numpy(), gd["d"], rtol=1e-6)
    assert np.allclose(o_loss.domain_loss(f, 1).numpy(), gd["t"], rtol=1e-6)


def test_pseudo_label_oracle_vs_golden():
    g = golden("pseudo_label")
    dets, _ = o_nms.non_max_suppression_ssod(g["pred"], g["thr"][0], g["thr"][1])
    t, invalid = o_pl.create_pseudo_label(dets, g["M_s"], int(g["hw"][1]), int(g["hw"][0]))
    assert not invalid and np.array_equal(t, g["targets"])
    t0, inv0 = o_pl.create_pseudo_label([np.zeros((0, 8), np.float32)] * 3, g["M_s"], 640, 640)
    assert inv0 and t0.shape == (0, 9)


def test_model_oracle_vs_golden():
    g = golden("model_tiny")
    m = o_model.Model(0.125, 0.33, 80)
    sd = {k[3:].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith("w__")}
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    m.eval()
    with torch.no_grad():
        (z, xs), feats = m(torch.from_numpy(g["x"]))
    assert np.allclose(z.numpy(), g["eval_z"], rtol=1e-5, atol=1e-5)
    m.train()
    pred, _ = m(torch.from_numpy(g["x"]))
    loss, _ = o_loss.compute_loss(pred, torch.from_numpy(g["targets"]), m.head.anchors, nc=80, box_w=0.05, obj_w=0.7,
                                  cls_w=0.3)
    assert np.allclose(loss.detach().numpy(), g["train_loss"], rtol=1e-5)
