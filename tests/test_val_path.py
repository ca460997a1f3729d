"""the per-batch inference path of val.run (reference val.py:277-338) end to end --
uint8 batch -> folded-BN inference forward -> non_max_suppression(conf 0.001, iou 0.6, multi_label=True) -- against the
oracle model + the oracle general NMS (both pinned to the live reference: tests/test_oracle_golden.py) on the same weights."""
import numpy as np
import torch

from tests.conftest import golden
from tests.test_ssod_step import make_trainer


def test_val_batch_inference_matches_oracle(hip):
    from efficientteacher_amd.val import infer_batch
    from oracle import model as o_model, nms as o_nms
    cfg, t = make_trainer(hip)                       # tiny SSOD detector with the golden weights + detection-friendly biases
    model = t.ema.ema
    ref = o_model.Model.from_cfg(cfg)
    ref.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()}, strict=True)
    ref.eval()
    rng = np.random.default_rng(8)
    img = torch.from_numpy(rng.integers(0, 256, (2, 3, 64, 64), dtype=np.uint8))
    dets, train_out = infer_batch(model, hip.t(img), conf_thres=0.25, iou_thres=0.6, half=False)
    with torch.no_grad():
        (z, _), _ = ref(img.float() / 255.0)
    want = o_nms.non_max_suppression(z.numpy(), 0.25, 0.6, multi_label=True)
    want = want[0] if isinstance(want, tuple) else want
    assert len(dets) == 2 and train_out is not None and len(train_out) == 3
    for got, w in zip(dets, want):
        got = got.cpu().numpy()
        assert got.shape == w.shape, (got.shape, w.shape)
        assert np.array_equal(got[:, 5], w[:, 5])                          # classes, in the same (score) order
        assert np.allclose(got[:, :4], w[:, :4], rtol=0, atol=2e-2)        # boxes in pixels
        assert np.allclose(got[:, 4], w[:, 4], rtol=1e-4, atol=1e-5)


def test_model_half_is_bf16_compute(hip):
    """val.py:212 calls model.half(): the compute dtype becomes bf16, the master weights stay fp32"""
    cfg, t = make_trainer(hip)
    m = t.ema.ema
    m.half()
    assert m._compute_dtype == torch.bfloat16 and next(m.parameters()).dtype == torch.float32
    m.float()
    assert m._compute_dtype == torch.float32
