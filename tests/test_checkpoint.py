"""Checkpoint interchange.  The state-dict round trip runs everywhere; the test against the
reference's own pickled checkpoint format needs the reference tree and therefore only runs in the build container
(it is skipped on the GPU box, where /root/reference does not exist)."""
import os

import pytest
import torch

from tests.conftest import ROOT

YAML = "efficientteacher_amd/configs/ssod/coco-standard/yolov5l_coco_ssod_10_percent.yaml"
TINY = ["Model.width_multiple", 0.125, "Model.depth_multiple", 0.33]


def _cfg():
    from efficientteacher_amd.configs import get_cfg
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, YAML))
    cfg.merge_from_list(TINY)
    return cfg


def test_state_dict_checkpoint_round_trip(tmp_path):
    from efficientteacher_amd.models.detector.yolo_ssod import Model
    from efficientteacher_amd.utils.checkpoint import load_reference_checkpoint, save_checkpoint
    torch.manual_seed(1)
    a = Model(_cfg())
    with torch.no_grad():
        for p in a.parameters():
            p.add_(torch.randn_like(p) * 0.01)
    p = str(tmp_path / "last.pt")
    save_checkpoint(p, a, epoch=7, best_fitness=0.5)
    torch.manual_seed(2)
    b = Model(_cfg())
    ck = load_reference_checkpoint(p, b)
    assert ck["epoch"] == 7 and ck["format"] == "state_dict"
    for (k, va), vb in zip(a.state_dict().items(), b.state_dict().values()):
        if va.is_floating_point():
            assert torch.allclose(va.half().float(), vb, rtol=0, atol=0), k      # stored as fp16 like the reference
        else:
            assert torch.equal(va, vb), k


def test_interchange_with_the_reference_format(tmp_path):
    from oracle import ref_loader
    if not ref_loader.available():
        pytest.skip("reference tree not present (GPU box)")
    ref_loader.load()
    from models.detector.yolo_ssod import Model as RefModel            # the reference's own class
    from configs.defaults import get_cfg as ref_get_cfg
    from efficientteacher_amd.models.detector.yolo_ssod import Model
    from efficientteacher_amd.utils.checkpoint import load_reference_checkpoint, save_checkpoint
    rcfg = ref_get_cfg()
    rcfg.merge_from_file(os.path.join(ref_loader.REF, "configs", "ssod", "coco-standard", "yolov5l_coco_ssod_10_percent.yaml"))
    rcfg.merge_from_list(TINY)
    torch.manual_seed(3)
    ref = RefModel(rcfg)
    # 1. a checkpoint written the way the reference writes it (trainer/trainer.py:475-481) loads into our Model
    import copy
    p1 = str(tmp_path / "ref_last.pt")
    torch.save({"epoch": 3, "best_fitness": 0.1, "model": copy.deepcopy(ref).half(), "ema": copy.deepcopy(ref).half(),
                "updates": 42, "optimizer": None, "wandb_id": None}, p1)
    mine = Model(_cfg())

    class _E:
        pass
    ema = _E(); ema.ema = Model(_cfg()); ema.updates = 0
    ck = load_reference_checkpoint(p1, mine, ema)
    assert ck["epoch"] == 3 and ema.updates == 42
    rsd = ref.half().float().state_dict()
    for k, v in mine.state_dict().items():
        assert torch.equal(v, rsd[k]) if not v.is_floating_point() else torch.allclose(v, rsd[k], rtol=0, atol=0), k
    # 2. a checkpoint written by save_checkpoint with the reference factory is what the reference resumes from
    with torch.no_grad():
        for q in mine.parameters():
            q.add_(0.5)
    p2 = str(tmp_path / "ours_last.pt")
    save_checkpoint(p2, mine, ema=ema, epoch=9, reference_model_factory=lambda: RefModel(rcfg))
    back = torch.load(p2, map_location="cpu", weights_only=False)
    assert isinstance(back["model"], RefModel) and back["epoch"] == 9
    csd = back["model"].float().state_dict()                           # reference trainer.py:135
    for k, v in mine.state_dict().items():
        if v.is_floating_point():
            assert torch.allclose(v.half().float(), csd[k], rtol=0, atol=0), k


def test_model_pickles_without_the_reference(hip, monkeypatch):
    """torch.save(model) where no reference tree is importable: the pickle carries (cfg, state_dict) and rebuilds this
    package's Model; with .half() the tensors are stored fp16 as the reference's checkpoints are"""
    import io
    from efficientteacher_amd.utils import checkpoint as ck
    from tests.test_model import build
    monkeypatch.setattr(ck, "_reference_model_class", lambda m: None)
    cfg, model, _ = build(hip, torch.float32)
    buf = io.BytesIO()
    torch.save({"model": model}, buf)
    buf.seek(0)
    m2 = torch.load(buf, map_location="cpu", weights_only=False)["model"]
    assert type(m2) is type(model)
    for k, v in model.state_dict().items():
        assert torch.equal(v.cpu(), m2.state_dict()[k]), k
