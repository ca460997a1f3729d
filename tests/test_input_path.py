"""the loaders' uint8 batches go to the model as they are -- normalisation inside the input pack
kernel (et_pack_input_u8), staging on a copy stream one step ahead (utils/prefetch.py)."""
import numpy as np
import pytest
import torch

from tests.conftest import golden
from tests.test_ssod_step import make_trainer


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hw", [(20, 24), (5, 3)], ids=["four pixels per thread (bf16)", "one pixel per thread"])
def test_pack_input_uint8_equals_float_division(hip, hw, dtype):
    """(float)x / 255 in the kernel is bit-identical to the IEEE division `imgs.float() / 255.0` (what torch computes on the CPU --
    the oracle's arithmetic; torch's GPU kernel multiplies by the rounded reciprocal instead, 1 ulp away) followed by the fp32 pack"""
    from efficientteacher_amd import ops
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.integers(0, 256, (2, 3) + hw, dtype=np.uint8))
    a = ops.pack_input(hip.t(x), dtype)
    b = ops.pack_input(hip.t(x.float() / 255.0), dtype)
    assert a.shape == (2,) + hw + (8,) and torch.equal(a.cpu(), b.cpu())
    ref = (x.float() / 255.0).permute(0, 2, 3, 1).to(dtype)
    assert torch.equal(a[..., :3].cpu(), ref)
    assert (a[..., 3:] == 0).all()


def test_train_with_unlabeled_on_loader_tuples(hip):
    """the reference's batch tuples (uint8 images, labels, paths, shapes[, imgs_ori, M_s]) through train_with_unlabeled: the
    losses of the first step equal a direct train_instance on the pre-normalised float tensors"""
    g = golden("ssod_step")
    u8 = lambda a: torch.from_numpy(np.round(a * 255).astype(np.uint8))
    imgs, u_str, u_ori = u8(g["imgs"]), u8(g["u_str"]), u8(g["u_ori"])
    targets, M_s = torch.from_numpy(g["targets"]), torch.from_numpy(g["M_s"])
    cfg, t1 = make_trainer(hip)
    out1 = t1.train_with_unlabeled([(imgs, targets, ["a", "b"], None)], [(u_str, None, ["c", "d"], None, u_ori, M_s)], start_ni=500)
    cfg, t2 = make_trainer(hip)
    f = lambda x: hip.t(x).float() / 255.0
    out2 = t2.train_instance(f(imgs), hip.t(targets), None, f(u_str), f(u_ori), None, hip.t(M_s), 500)
    for k in out2:
        assert abs(float(out1[k]) - float(out2[k])) <= 1e-6 * max(1.0, abs(float(out2[k]))), k


def test_train_with_unlabeled_generates_the_strong_view(hip):
    """an unlabeled batch that carries the weak view only (imgs = None, M_s = None): the strong view and its M_s rows come from
    utils/augment.StrongViewGenerator on the device, and the step runs on them"""
    g = golden("ssod_step")
    u8 = lambda a: torch.from_numpy(np.round(a * 255).astype(np.uint8))
    imgs, u_ori = u8(g["imgs"]), u8(g["u_ori"])
    targets = torch.from_numpy(g["targets"])
    cfg, t = make_trainer(hip)
    out = t.train_with_unlabeled([(imgs, targets, ["a", "b"], None)], [(None, None, ["c", "d"], None, u_ori, None)], start_ni=500)
    assert t._strong_view is not None
    assert all(np.isfinite(float(v)) for v in out.values()) and {"ss_box", "ss_obj", "ss_cls"} <= set(out)


@pytest.mark.gpu
def test_prefetcher_keeps_consecutive_batches_apart():
    """equal-shaped consecutive batches shared one pinned slot and were overwritten while their asynchronous
    host->device copy was still reading it.  Eight distinct uint8 batches (+ labels of varying length) through the
    prefetcher: every delivered batch equals its source, and the pinned arenas stay bounded (one per item and slot)."""
    from efficientteacher_amd.utils.prefetch import DevicePrefetcher
    if not torch.cuda.is_available():
        pytest.fail("-m gpu selected but no GPU is visible")
    rng = np.random.default_rng(0)
    src = [(torch.from_numpy(rng.integers(0, 256, (8, 3, 320, 320), dtype=np.uint8)),
            torch.from_numpy(rng.random((3 + i, 6), dtype=np.float32)), ["p"] * 8) for i in range(8)]
    pf = DevicePrefetcher(iter(src), "cuda:0")
    got = []
    for b in pf:
        torch.cuda._sleep(20_000_000)                    # keep the compute stream busy: the copy stream runs ahead
        got.append((b[0].clone(), b[1].clone()))
    torch.cuda.synchronize()
    assert len(got) == len(src)
    for (gi, gl), (si, sl, _) in zip(got, src):
        assert torch.equal(gi.cpu(), si) and torch.equal(gl.cpu(), sl)
    assert len(pf._pinned) <= 2 * (pf.depth + 1)


def test_prefetcher_cpu_passthrough():
    from efficientteacher_amd.utils.prefetch import DevicePrefetcher
    src = [(torch.full((2, 3), i), None, "x") for i in range(4)]
    out = list(DevicePrefetcher(iter(src), "cpu"))
    assert len(out) == 4 and all(torch.equal(o[0], s[0]) and o[2] == "x" for o, s in zip(out, src))
