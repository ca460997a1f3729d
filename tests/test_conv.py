"""Implicit-GEMM conv kernels (fwd / dgrad / wgrad) vs a plain PyTorch fp32 reference of the same op.

fp32 (parity mode, exact-f32 MFMA): tolerance 2e-5 relative to the output scale.
bf16 (performance mode): inputs are rounded to bf16 first, the reference runs in fp32 on those rounded
values, tolerance 2e-2 (bf16 output rounding + accumulation order).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

CASES = [
    # N, H, W, Cin, Cout, k, s, p
    (2, 12, 12, 32, 64, 1, 1, 0),
    (1, 9, 11, 64, 40, 3, 1, 1),      # ragged M, Cout guard, BKV=8
    (2, 12, 12, 16, 128, 3, 2, 1),    # stride 2, BKV=4 (fp32) / non-uniform tap (bf16: CV=2)
    (1, 16, 16, 8, 32, 6, 2, 2),      # the stem: Cin padded to 8, 36 taps
    (1, 6, 6, 128, 136, 1, 1, 0),     # wide N tile with a ragged second tile
    (1, 5, 5, 32, 264, 3, 1, 1),      # Cout and Cin*taps beyond 256: the 256-wide wgrad tiles, ragged
]


def _mk(hip, shape, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.randn(shape, generator=g)
    return t.to(dtype).to(hip.device)


def _ref_conv(x, w, s, p):
    # x (N,H,W,C), w (Cout,KH,KW,Cin) -> NHWC fp32
    y = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.float().cpu().permute(0, 3, 1, 2), stride=s, padding=p)
    return y.permute(0, 2, 3, 1).contiguous()


def _tol(dtype):
    return 2e-5 if dtype == torch.float32 else 2e-2


DTYPES = [torch.float32, torch.bfloat16, torch.float16]      # parity mode, performance mode, the reference's AMP arithmetic (r05)


def kn(*a, **k):
    """the library's kernel name (et_conv2d_kernel_name: what rocprofv3 prints) with the storage-type template argument removed --
    `conv_gemm_rs_kernel<unsigned short, 128, ...>` / `<et_f16, 128, ...>` -> `conv_gemm_rs_kernel<128, ...>`,
    `conv_gemm_pprs_kernel<et_f16>` -> `conv_gemm_pprs_kernel`: the tile selection pinned below does not depend on the 16-bit
    format, and one table serves both (the type itself is asserted in test_kernel_names_carry_the_storage_type)"""
    from efficientteacher_amd import ops
    n = ops.kernel_name(*a, **k)
    for t in ("unsigned short", "et_f16", "float"):
        n = n.replace(f"<{t}>", "").replace(f"<{t}, ", "<")
    return n


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CASES)
def test_conv_fwd(hip, case, dtype):
    from efficientteacher_amd import ops
    N, H, W, Cin, Cout, k, s, p = case
    x = _mk(hip, (N, H, W, Cin), dtype, 1)
    w = _mk(hip, (Cout, k, k, Cin), dtype, 2) * (1.0 / (k * k * Cin) ** 0.5)
    w = w.to(dtype)
    y, stats = ops.conv2d_fwd(x, w, s, p, want_stats=True)
    ref = _ref_conv(x, w, s, p)
    err = (y.float().cpu() - ref).abs().max().item()
    assert err <= _tol(dtype) * max(1.0, ref.abs().max().item()), err
    # BN partial statistics: sum / sum of squares over pixels of the raw accumulators
    st = stats.sum(0).cpu()
    flat = ref.reshape(-1, Cout)
    assert torch.allclose(st[0], flat.sum(0), rtol=1e-3, atol=_tol(dtype) * flat.shape[0] ** 0.5 * 4)
    assert torch.allclose(st[1], (flat ** 2).sum(0), rtol=2e-2 if dtype != torch.float32 else 1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_fwd_epilogue_and_slices(hip, dtype):
    """bias + SiLU + residual, reading a channel slice and writing into a slice of a wider buffer."""
    from efficientteacher_amd import ops
    N, H, W, Cin, Cout = 2, 8, 8, 32, 24
    xb = _mk(hip, (N, H, W, 64), dtype, 3)
    x = xb[..., 16:48]
    w = (_mk(hip, (Cout, 3, 3, Cin), dtype, 4) * 0.08).to(dtype)
    bias = _mk(hip, (Cout,), torch.float32, 5)
    res = _mk(hip, (N, H, W, Cout), dtype, 6)
    outb = torch.zeros((N, H, W, 40), dtype=dtype, device=hip.device)
    out = outb[..., 8:32]
    ops.conv2d_fwd(x, w, 1, 1, bias=bias, act=ops.ACT_SILU, residual=res, out=out)
    ref = _ref_conv(x, w, 1, 1) + bias.cpu()
    ref = F.silu(ref) + res.float().cpu()
    err = (out.float().cpu() - ref).abs().max().item()
    assert err <= _tol(dtype) * 8, err
    assert (outb[..., :8] == 0).all() and (outb[..., 32:] == 0).all()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [c for c in CASES if c[3] % 8 == 0 and c[4] % 8 == 0 and c[5] != 6])
def test_conv_dgrad(hip, case, dtype):
    from efficientteacher_amd import ops
    N, H, W, Cin, Cout, k, s, p = case
    OH, OW = ops.conv_out_hw(H, W, k, s, p)
    dy = _mk(hip, (N, OH, OW, Cout), dtype, 7)
    w = (_mk(hip, (Cout, k, k, Cin), dtype, 8) * (1.0 / (k * k * Cout) ** 0.5)).to(dtype)
    wT = ops.weight_transpose(w)
    assert torch.equal(wT.cpu(), w.cpu().permute(3, 1, 2, 0).contiguous())
    dx = ops.conv2d_dgrad(dy, wT, (H, W), s, p)
    xr = torch.zeros((N, Cin, H, W), requires_grad=True)
    yr = F.conv2d(xr, w.float().cpu().permute(0, 3, 1, 2), stride=s, padding=p)
    yr.backward(dy.float().cpu().permute(0, 3, 1, 2))
    ref = xr.grad.permute(0, 2, 3, 1)
    err = (dx.float().cpu() - ref).abs().max().item()
    assert err <= _tol(dtype) * max(1.0, ref.abs().max().item()), err
    # accumulate form
    dx2 = dx.clone()
    ops.conv2d_dgrad(dy, wT, (H, W), s, p, out=dx2, accumulate=True)
    assert torch.allclose(dx2.float().cpu(), 2 * dx.float().cpu(), rtol=2e-2, atol=1e-2 if dtype != torch.float32 else 1e-5)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 9, 11, 256), (1, 7, 5, 24), (3, 20, 20, 64)])
def test_colsum_bias_gradient(hip, shape, dtype):
    """et_colsum: out[c] += sum over pixels (the bias gradient of the Detect convs), on a channel slice of a wider buffer; bf16 with
    whole 8-channel vectors takes the 16-byte-load kernel, 24 channels (3 vectors: not a divisor of 256) the element-wise one"""
    from efficientteacher_amd import ops
    N, H, W, C = shape
    wide = _mk(hip, (N, H, W, C + 16), dtype, 71)
    x = wide[..., 8:8 + C]
    out = torch.full((C,), 0.5, dtype=torch.float32, device=hip.device)
    ops.colsum(x, out)
    ref = x.float().cpu().reshape(-1, C).sum(0) + 0.5
    assert torch.allclose(out.cpu(), ref, rtol=1e-4, atol=1e-3 * (N * H * W) ** 0.5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_weight_transpose_all_layers(hip, dtype):
    """et_weight_transpose_all: every layer of a flat weight arena [Cout][KH][KW][Cin] -> [Cin][KH][KW][Cout] in one launch
    (bf16: the tiled 8x8-register-block kernel; fp32: one element per thread), ragged 64-tiles, gaps between layers untouched"""
    from efficientteacher_amd import ops
    layers = [(72, 9, 40), (8, 1, 136), (128, 9, 64), (200, 1, 8), (20, 4, 12)]   # (Cout, taps, Cin); the last one is not whole 16-byte rows
    offs, total = [], 0
    for co, tt, ci in layers:
        offs.append(total)
        total += (co * tt * ci + 15) // 16 * 16 + 16                          # 16-element alignment + a gap
    g = torch.Generator().manual_seed(5)
    arena = torch.randn(total, generator=g).to(dtype).to(hip.device)
    out = torch.full((total,), 7.0, dtype=dtype, device=hip.device)
    table = torch.tensor([[o, co, tt, ci] for o, (co, tt, ci) in zip(offs, layers)], dtype=torch.int32, device=hip.device)
    ops.weight_transpose_all(arena, out, table, total)
    for o, (co, tt, ci) in zip(offs, layers):
        n = co * tt * ci
        ref = arena[o:o + n].view(co, tt, ci).permute(2, 1, 0).contiguous().view(-1)
        assert torch.equal(out[o:o + n].cpu(), ref.cpu()), (co, tt, ci)
        assert (out[o + n:o + n + 16].float().cpu() == 7.0).all()           # the gap behind the layer is not written


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [c for c in CASES if c[4] % 8 == 0])
def test_conv_wgrad(hip, case, dtype):
    from efficientteacher_amd import ops
    N, H, W, Cin, Cout, k, s, p = case
    OH, OW = ops.conv_out_hw(H, W, k, s, p)
    x = _mk(hip, (N, H, W, Cin), dtype, 9)
    dy = _mk(hip, (N, OH, OW, Cout), dtype, 10)
    dw = torch.zeros((Cout, k, k, Cin), dtype=torch.float32, device=hip.device)
    ops.conv2d_wgrad(x, dy, dw, k, s, p)
    wr = torch.zeros((Cout, Cin, k, k), requires_grad=True)
    yr = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), wr, stride=s, padding=p)
    yr.backward(dy.float().cpu().permute(0, 3, 1, 2))
    ref = wr.grad.permute(0, 2, 3, 1)
    err = (dw.cpu() - ref).abs().max().item()
    assert err <= 1e-4 * max(1.0, ref.abs().max().item()), err
    ops.conv2d_wgrad(x, dy, dw, k, s, p)          # accumulates
    assert torch.allclose(dw.cpu(), 2 * ref, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_wgrad_grouped(hip, dtype):
    """et_conv2d_wgrad_grouped: several same-shaped layers in one launch, one of them reading a channel slice of a
    wider buffer (different pixel stride) -- each dW must equal its own single-layer gradient."""
    from efficientteacher_amd import ops
    N, H, W, Cin, Cout, k, s, p = 2, 10, 10, 32, 48, 3, 1, 1
    items, refs = [], []
    for i in range(3):
        if i == 1:
            wide = _mk(hip, (N, H, W, Cin + 16), dtype, 20 + i)
            x = wide[..., 8:8 + Cin]
        else:
            x = _mk(hip, (N, H, W, Cin), dtype, 20 + i)
        dy = _mk(hip, (N, H, W, Cout), dtype, 30 + i)
        dw = torch.zeros((Cout, k, k, Cin), dtype=torch.float32, device=hip.device)
        items.append((x, dy, dw))
        wr = torch.zeros((Cout, Cin, k, k), requires_grad=True)
        F.conv2d(x.float().cpu().permute(0, 3, 1, 2), wr, stride=s, padding=p).backward(dy.float().cpu().permute(0, 3, 1, 2))
        refs.append(wr.grad.permute(0, 2, 3, 1))
    ops.conv2d_wgrad_grouped(items, k, s, p)
    for (_, _, dw), ref in zip(items, refs):
        err = (dw.cpu() - ref).abs().max().item()
        assert err <= 1e-4 * max(1.0, ref.abs().max().item()), err


# ---- every kernel instantiation the YOLOv5l / YOLOv5s bench launches, selected on purpose ----------------------
# (VERDICT r01 "parity gap": the bench-dominant tiles were never compared element-wise with anything.)  Each case
# names the instantiation csrc/conv.hip must pick for it -- et_conv2d_kernel_name is the selection logic itself, not
# a copy -- so a change of the tile policy that silently drops a kernel out of test coverage fails here.
GLDS = "conv_gemm_glds_kernel<"
RS128, RS64 = "conv_gemm_rs_kernel<128, 128, 2, 2>", "conv_gemm_rs_kernel<128, 64, 2, 2>"
S1 = "conv1x1_stream_kernel<"          # prefix: the shape table behind it (csrc/conv.hip plan_s1) is pinned by test_stream_kernel_shape_table
SELECT = [
    # N, H, W, Cin, Cout, k, s, p, fwd kernel, dgrad kernels (per parity class), wgrad kernel
    ((2, 20, 20, 256, 256, 3, 1, 1), "conv_gemm_pprs_kernel", ["conv_gemm_pprs_kernel"], "conv_wgrad_rs_kernel<128, 128, 2, 4>"),
    ((2, 20, 20, 128, 256, 3, 2, 1), "conv_gemm_pp_kernel",
     [GLDS + "128, 128, 2, 2, 4, 3, true>", GLDS + "128, 128, 2, 2, 8, 2, true>", GLDS + "128, 128, 2, 2, 8, 2, true>",
      GLDS + "128, 128, 2, 2, 8, 2, true>"], "conv_wgrad_tr_kernel<256, 256, 2, 4>"),
    ((2, 20, 20, 512, 512, 1, 1, 0), GLDS + "128, 128, 2, 2, 8, 2, true>", [GLDS + "128, 128, 2, 2, 8, 2, true>"],
     "conv_wgrad_tr_kernel<256, 256, 2, 4>"),                                                  # r06: >= 512-channel 1x1 layers on the 256^2 tile (identity X rows)
    ((1, 9, 11, 64, 40, 3, 1, 1), RS64, [GLDS + "128, 64, 2, 2, 4, 2, false>"],                # ragged M, 11-pixel rows
     "conv_wgrad_rs_kernel<64, 64, 2, 2>"),
    ((2, 12, 12, 64, 128, 1, 1, 0), S1, [S1], "conv_wgrad_tr_kernel<128, 64, 2, 2>"),
    ((2, 12, 12, 128, 128, 3, 1, 1), RS128, [RS128],
     "conv_wgrad_rs_kernel<128, 128, 2, 4>"),
    ((2, 12, 12, 64, 64, 1, 1, 0), S1, [S1], "conv_wgrad_tr_kernel<64, 64, 2, 2>"),
    ((1, 16, 16, 8, 32, 6, 2, 2), "conv_stem_kernel", None, None),                            # the stem (no dgrad in the net)
    ((1, 5, 5, 64, 264, 3, 1, 1), "conv_gemm_pprs_kernel", [GLDS + "128, 64, 2, 2, 4, 2, false>"],
     "conv_wgrad_tr_kernel<256, 256, 2, 4>"),                                                  # ragged M and Cout on the 256^2 tiles
    ((3, 14, 14, 192, 320, 3, 1, 1), "conv_gemm_pprs_kernel", [RS128], "conv_wgrad_rs_kernel<128, 128, 2, 4>"),       # ragged cout / cin tiles
    # YOLOv8 head: its 68 (-> 72) channel DFL branch is not a multiple of the 32-wide K chunk: per-lane tap decode (UTAP false)
    ((1, 9, 11, 256, 72, 3, 1, 1), RS128, [GLDS + "128, 128, 2, 2, 4, 2, false>"],
     "conv_wgrad_tr_kernel<128, 128, 2, 2>"),
    # YOLOv5s (BASELINE configs[1]): 32 -> 64 stride-2 3x3, K = 288 in 32-wide chunks on the 64-wide tile
    ((2, 24, 24, 32, 64, 3, 2, 1), GLDS + "128, 64, 2, 2, 4, 2, true>", [GLDS + "128, 64, 2, 2, 4, 3, true>"] * 4,
     "conv_wgrad_tr_kernel<64, 128, 2, 2>"),
    # 64 -> 128 stride-2 3x3 (the layer after the stem): its four-tap dgrad class is the 64-wide tile with 64-wide chunks
    ((2, 24, 24, 64, 128, 3, 2, 1), GLDS + "128, 128, 2, 2, 8, 2, true>",
     [GLDS + "128, 64, 2, 2, 4, 3, true>"] * 3 + [GLDS + "128, 64, 2, 2, 8, 2, true>"], "conv_wgrad_rs_kernel<128, 64, 2, 2, 2>"),
    # ... an ODD input size: the four parity classes differ in pixel count (1, 2, 2, 4 taps: K = 128, 256, 256, 512)
    ((1, 15, 15, 64, 128, 3, 2, 1), GLDS + "128, 128, 2, 2, 8, 2, true>",
     [GLDS + "128, 64, 2, 2, 4, 3, true>"] * 3 + [GLDS + "128, 64, 2, 2, 8, 2, true>"], "conv_wgrad_tr_kernel<128, 128, 2, 2>"),
    # the single-launch form of the short-K ring on the 128-wide tile: dgrad of a 512 -> 256 1x1 layer (K = 256, 512 dX channels)
    ((1, 12, 12, 512, 256, 1, 1, 0), GLDS + "128, 128, 2, 2, 8, 2, true>", [GLDS + "128, 128, 2, 2, 4, 3, true>"], "conv_wgrad_tr_kernel<128, 128, 2, 2>"),
    # ... and >= 256 dX channels: the classes run on the 256x256 ping-pong tile, one launch each
    ((1, 16, 16, 256, 512, 3, 2, 1), "conv_gemm_pp_kernel", ["conv_gemm_pp_kernel"] * 4, "conv_wgrad_tr_kernel<256, 256, 2, 4>"),
    # the persistent streaming kernel of the 1x1 layers with K <= 256 (every instantiation; ragged M, Cout below the column tile,
    # and -- last two -- a channel count whose dgrad is not eligible, i.e. a tiled dgrad beside a streamed forward)
    ((2, 13, 11, 64, 64, 1, 1, 0), S1, [S1], "conv_wgrad_tr_kernel<64, 64, 2, 2>"),
    ((1, 17, 19, 128, 128, 1, 1, 0), S1, [S1], "conv_wgrad_tr_kernel<128, 128, 2, 2>"),
    ((2, 9, 10, 256, 256, 1, 1, 0), S1, [S1], "conv_wgrad_tr_kernel<128, 128, 2, 2>"),
    ((1, 12, 12, 256, 128, 1, 1, 0), S1, [S1], "conv_wgrad_tr_kernel<128, 128, 2, 2>"),
    ((1, 12, 12, 128, 256, 1, 1, 0), S1, [S1], "conv_wgrad_tr_kernel<128, 128, 2, 2>"),
    ((2, 7, 9, 128, 64, 1, 1, 0), S1, [S1], "conv_wgrad_tr_kernel<64, 128, 2, 2>"),
    ((1, 9, 9, 256, 200, 1, 1, 0), S1, [GLDS + "128, 128, 2, 2, 4, 2, false>"], "conv_wgrad_tr_kernel<128, 128, 2, 2>"),
    ((1, 8, 8, 64, 24, 1, 1, 0), S1, [GLDS + "128, 64, 2, 2, 4, 2, false>"], "conv_wgrad_tr_kernel<64, 64, 2, 2>"),
    ((1, 10, 11, 64, 256, 1, 1, 0), S1, [GLDS + "128, 64, 2, 2, 4, 3, true>"], "conv_wgrad_tr_kernel<128, 64, 2, 2>"),
]
STREAM_CASES = [c for c in SELECT if c[1].startswith(S1)]
PPRS_CASES = [c for c in SELECT if c[1] in ("conv_gemm_pprs_kernel", RS128, RS64) or (c[2] and any(k in ("conv_gemm_pprs_kernel", RS128, RS64) for k in c[2]))]


@pytest.mark.parametrize("case,kf,kd,kw", SELECT, ids=[str(c[0]) for c in SELECT])
def test_bench_instantiations_elementwise(hip, case, kf, kd, kw):
    """bf16 fwd (+ stats, scale/bias/SiLU/residual epilogue, output slice), dgrad (+ residual, accumulate) and wgrad of
    the named instantiation, element-wise against F.conv2d (fp32) on the bf16-rounded operands."""
    _check_instantiation(hip, case, kf, kd, kw)


@pytest.mark.parametrize("case,kf,kd,kw", PPRS_CASES, ids=[str(c[0]) for c in PPRS_CASES])
def test_row_shift_flat_address_twins(hip, case, kf, kd, kw, monkeypatch):
    """conv_gemm_pprs_kernel / conv_gemm_rs_kernel stage their LDS-DMA pieces through buffer descriptors (out-of-range lanes land as
    zeros: no zero page); conv_gemm_pprs_flat_kernel / conv_gemm_rs_flat_kernel are the same loops on flat 64-bit addresses, taken for
    an operand of 2^31 bytes or more and under ET_CONV_BUF_DMA=0: the same element-wise checks on that arm.  (The buffer form is what every
    other test of these shapes runs; the full-size idempotence / linearity tests of tests/test_fullsize.py are the ones that caught the
    LDS-ring race it exposed, profiles/r06_lds_ring_war_race.txt.)"""
    assert len(PPRS_CASES) >= 6
    monkeypatch.setenv("ET_CONV_BUF_DMA", "0")
    _check_instantiation(hip, case, kf, kd, kw)


FLAT_TWIN_CASES = [c for c in STREAM_CASES] + [c for c in SELECT if c[0][5] == 1 and c[0][6] == 1 and c not in STREAM_CASES][:4]


@pytest.mark.parametrize("case,kf,kd,kw", FLAT_TWIN_CASES, ids=[str(c[0]) for c in FLAT_TWIN_CASES])
def test_stream_and_1x1_wgrad_flat_address_twins(hip, case, kf, kd, kw, monkeypatch):
    """conv1x1_stream_kernel and the 1x1 stride-1 arm of conv_wgrad_tr_kernel stage through buffer descriptors too (r06); under
    ET_CONV_BUF_DMA=0 (and for tensors of 2^31 bytes or more) conv1x1_stream_flat_kernel / the flat pieces of the weight gradient run:
    the same element-wise checks on that arm."""
    assert len(STREAM_CASES) >= 4 and len(FLAT_TWIN_CASES) > len(STREAM_CASES)
    monkeypatch.setenv("ET_CONV_BUF_DMA", "0")
    _check_instantiation(hip, case, kf, kd, kw)


@pytest.mark.parametrize("case,kf,kd,kw", SELECT, ids=[str(c[0]) for c in SELECT])
def test_bench_instantiations_elementwise_fp16(hip, case, kf, kd, kw):
    """the same instantiations with T = et_f16 (IEEE half, v_mfma_f32_32x32x16_f16: the reference's AMP arithmetic, r05): the tile
    selection is the bf16 one, the element-wise comparison is against F.conv2d (fp32) on the fp16-rounded operands"""
    N, H, W, Cin, Cout, k = case[:6]
    if hip.emulated and N * H * W * Cin * Cout * k * k > 6e7:
        pytest.skip("CPU emulator tier: the small cases cover every kernel family in fp16 (the bf16 table above runs all sizes); "
                    "the GPU tier runs every case in fp16 too")
    _check_instantiation(hip, case, kf, kd, kw, dt=torch.float16)


def test_kernel_names_carry_the_storage_type(hip):
    """what rocprofv3 prints: the 16-bit kernels are templates on the storage type (unsigned short = bf16, et_f16 = IEEE half)"""
    from efficientteacher_amd import ops
    a = (2, 20, 20, 256, 256, 3, 1, 1)
    assert ops.kernel_name("fwd", torch.bfloat16, *a) == "conv_gemm_pprs_kernel<unsigned short>"
    assert ops.kernel_name("fwd", torch.float16, *a) == "conv_gemm_pprs_kernel<et_f16>"
    assert ops.kernel_name("wgrad", torch.float16, *a) == "conv_wgrad_rs_kernel<et_f16, 128, 128, 2, 4>"
    assert ops.kernel_name("fwd", torch.float16, 2, 12, 12, 128, 128, 3, 1, 1) == "conv_gemm_rs_kernel<et_f16, 128, 128, 2, 2>"
    assert ops.kernel_name("fwd", torch.bfloat16, 2, 12, 12, 64, 64, 1, 1, 0).startswith("conv1x1_stream_kernel<unsigned short, 1, ")
    assert ops.kernel_name("fwd", torch.float32, 2, 12, 12, 64, 64, 1, 1, 0).startswith("conv_gemm_glds_kernel<float, ")


@pytest.mark.parametrize("dma_late,seed", [(1, 3), (0, 5), (1, 11)])
def test_lds_dma_pipelines_under_adversarial_schedules(emu, dma_late, seed):
    """The counted-vmcnt LDS-DMA pipelines (the ping-pong 256x256 gather-GEMM with per-tap and with shared activation rows, the 3-deep
    short-K ring, the two-ring row-shift 3x3 kernel, the wgrad double buffer) under the emulator's race-exposing modes: DMA landing as late / as early as the hardware may, waves run
    one at a time in random order between barriers.  A wait that is one half-tile too weak, or a half-tile staged
    into a buffer that is still being read, fails here (checked by mutation when the kernel was written)."""
    emu.configure(dma_late, seed)
    for case, kf, kd, kw in (SELECT[0], SELECT[1], SELECT[4], SELECT[8], SELECT[3], SELECT[5]):     # [3], [5]: the row-shift 3x3 tiles; [1]: stride-2 wgrad
        _check_instantiation(emu, case, kf, kd, kw)
    import os
    os.environ["ET_CONV_STEM_WGS"] = "2"          # 6 tiles on 2 persistent workgroups: the single patch buffer is re-staged
    try:
        _stem_case(emu, 1, 20, 300, 48)
    finally:
        del os.environ["ET_CONV_STEM_WGS"]


@pytest.mark.parametrize("wgs", [1, 3])
@pytest.mark.parametrize("dma_late,seed", [(1, 3), (0, 5), (1, 11)])
def test_stream_kernel_tile_loop_under_adversarial_schedules(emu, dma_late, seed, wgs, monkeypatch):
    """conv1x1_stream_kernel with a grid of 1 / 3 persistent workgroups: every workgroup walks several row tiles, the chunk ring
    wraps around across tile boundaries and the statistics accumulate over the tiles -- under the emulator's race-exposing modes (a
    counted wait that is one chunk too weak, or a chunk staged into a slot that is still being read, fails here)."""
    monkeypatch.setenv("ET_CONV_S1_WGS", str(wgs))
    emu.configure(dma_late, seed)
    for case, kf, kd, kw in STREAM_CASES:
        _check_instantiation(emu, case, kf, kd, kw)
    if seed == 3:        # the FULL epilogue (residual + BN-backward sums accumulated over the tiles of a workgroup), one schedule
        for case in ((2, 9, 11, 256, 256, 1), (1, 13, 13, 128, 128, 1), (2, 9, 9, 64, 64, 1)):
            test_dgrad_with_fused_bn_backward_sums(emu, case, torch.bfloat16)
    # the sharded statistics tail (r05): the wave rows of a workgroup meet in the chunk ring's LDS after the last tile -- a wave still
    # reading the ring, or a sum read before every wave has written its row, shows under these schedules
    from efficientteacher_amd import ops
    from efficientteacher_amd.flat_state import BN_SHARDS
    for (N, H, W, Cin, Cout) in ((2, 9, 11, 128, 128), (1, 13, 13, 64, 64), (2, 9, 9, 256, 256)):
        x = _mk(emu, (N, H, W, Cin), torch.bfloat16, 601)
        w = (_mk(emu, (Cout, 1, 1, Cin), torch.bfloat16, 602) * Cin ** -0.5).to(torch.bfloat16)
        assert kn("fwd", torch.bfloat16, N, H, W, Cin, Cout, 1, 1, 0).startswith("conv1x1_stream_kernel")
        y0, stats = ops.conv2d_fwd(x, w, 1, 0, want_stats=True)
        full = torch.zeros((BN_SHARDS, 2, Cout + 24), dtype=torch.float32)
        y1 = ops.conv2d_fwd(x, w, 1, 0, shards=(full.view(-1)[16:], Cout + 24))
        assert torch.equal(y0, y1)
        ref = stats.double().sum(0)
        assert torch.allclose(full.sum(0)[:, 16:16 + Cout].double(), ref, rtol=2e-5, atol=1e-4 * ref.abs().max().item())
        assert torch.count_nonzero(full[:, :, :16]) == 0 and torch.count_nonzero(full[:, :, 16 + Cout:]) == 0


def test_stream_kernel_statistics_rows_follow_the_grid(hip, monkeypatch):
    """the persistent kernel writes one partial row per (workgroup, row group): ops.stats_rows reports that count, every row is
    written (a buffer pre-filled with NaN sums to the reference), and the row count follows ET_CONV_S1_WGS"""
    from efficientteacher_amd import ops
    N, H, W, C = 2, 15, 15, 128
    dt = torch.bfloat16
    x = _mk(hip, (N, H, W, C), dt, 81)
    w = (_mk(hip, (C, 1, 1, C), dt, 82) * C ** -0.5).to(dt)
    ref = _ref_conv(x, w, 1, 0).reshape(-1, C)
    for wgs, rows in ((1, 2), (2, 4), (1000, 8)):        # 450 pixels = 4 row tiles of 128; two row groups (WM = 2) per workgroup
        monkeypatch.setenv("ET_CONV_S1_WGS", str(wgs))
        assert ops.stats_rows("fwd", dt, N, H, W, C, C, 1, 1, 0) == rows
        y, st = ops.conv2d_fwd(x, w, 1, 0, want_stats=True)
        assert st.shape == (rows, 2, C)
        assert torch.allclose(st.sum(0)[0].cpu(), ref.sum(0), rtol=1e-3, atol=0.5)
        assert torch.allclose(st.sum(0)[1].cpu(), (ref ** 2).sum(0), rtol=2e-2, atol=1e-3)


@pytest.mark.parametrize("C,K", [(256, 256), (128, 128), (64, 128)])
def test_forward_statistics_with_a_residual_use_the_full_plans_row_count(hip, monkeypatch, C, K):
    """et_conv2d_fwd(stats_partial + residual) runs the FULL-epilogue plan; for the persistent 1x1 kernel that
    plan has another tile height / grid than the plain one (K = 128 -> 128: 64-row instead of 128-row tiles, twice the rows), so the row count
    must be asked with op 'fwd_res'.  Every row is written (NaN-prefilled by torch.empty would poison the sums) and the sums are
    those of the raw accumulators; the output carries the residual."""
    from efficientteacher_amd import ops
    N, H, W = 2, 13, 11
    dt = torch.bfloat16
    x = _mk(hip, (N, H, W, K), dt, 91)
    w = (_mk(hip, (C, 1, 1, K), dt, 92) * K ** -0.5).to(dt)
    res = _mk(hip, (N, H, W, C), dt, 93)
    ref = _ref_conv(x, w, 1, 0)
    for wgs in (1, 3, 1000):
        monkeypatch.setenv("ET_CONV_S1_WGS", str(wgs))
        rows_plain = ops.stats_rows("fwd", dt, N, H, W, K, C, 1, 1, 0)
        rows_full = ops.stats_rows("fwd_res", dt, N, H, W, K, C, 1, 1, 0)
        assert kn("fwd_res", dt, N, H, W, K, C, 1, 1, 0).startswith(S1)
        y, st = ops.conv2d_fwd(x, w, 1, 0, residual=res, want_stats=True)
        assert st.shape == (rows_full, 2, C) and torch.isfinite(st).all()
        flat = ref.reshape(-1, C)
        assert torch.allclose(st.sum(0)[0].cpu(), flat.sum(0), rtol=1e-3, atol=0.5)
        assert torch.allclose(st.sum(0)[1].cpu(), (flat ** 2).sum(0), rtol=2e-2, atol=1e-3)
        assert (y.float().cpu() - (ref + res.float().cpu())).abs().max().item() <= 3e-2 * max(1.0, ref.abs().max().item())
        if wgs == 1000 and (C, K) == (128, 128):
            assert rows_full > rows_plain           # the plans differ in tile height (64 vs 128 rows): asking with "fwd" would under-allocate


def _same(name, want):
    """exact name, or -- for the persistent 1x1 kernel -- the kernel family (its template arguments are pinned elsewhere)"""
    return name.startswith(want) if want == S1 else name == want


def _check_instantiation(hip, case, kf, kd, kw, dt=torch.bfloat16):
    from efficientteacher_amd import ops
    N, H, W, Cin, Cout, k, s, p = case
    assert _same(kn("fwd", dt, N, H, W, Cin, Cout, k, s, p), kf)
    x = _mk(hip, (N, H, W, Cin), dt, 41)
    w = (_mk(hip, (Cout, k, k, Cin), dt, 42) * (1.0 / (k * k * Cin) ** 0.5)).to(dt)
    OH, OW = ops.conv_out_hw(H, W, k, s, p)
    ref = _ref_conv(x, w, s, p)
    y, stats = ops.conv2d_fwd(x, w, s, p, want_stats=True)
    scale_ref = max(1.0, ref.abs().max().item())
    assert (y.float().cpu() - ref).abs().max().item() <= 2e-2 * scale_ref
    flat = ref.reshape(-1, Cout)
    st = stats.sum(0).cpu()
    assert torch.allclose(st[0], flat.sum(0), rtol=1e-3, atol=2e-2 * flat.shape[0] ** 0.5 * 4)
    assert torch.allclose(st[1], (flat ** 2).sum(0), rtol=2e-2, atol=1e-3)
    # teacher-style epilogue: folded BN scale/bias + SiLU + residual, written into a channel slice of a wider buffer
    sc = _mk(hip, (Cout,), torch.float32, 43).abs() + 0.5
    bi = _mk(hip, (Cout,), torch.float32, 44)
    res = _mk(hip, (N, OH, OW, Cout), dt, 45)
    wide = torch.zeros((N, OH, OW, Cout + 16), dtype=dt, device=hip.device)
    ops.conv2d_fwd(x, w, s, p, scale=sc, bias=bi, act=ops.ACT_SILU, residual=res, out=wide[..., 8:8 + Cout])
    ref2 = F.silu(ref * sc.cpu() + bi.cpu()) + res.float().cpu()
    assert (wide[..., 8:8 + Cout].float().cpu() - ref2).abs().max().item() <= 3e-2 * max(1.0, ref2.abs().max().item())
    assert (wide[..., :8] == 0).all() and (wide[..., 8 + Cout:] == 0).all()
    if kd is None:
        return
    # dgrad (the operand roles swap: K = taps * Cout)
    names = [kn("dgrad", dt, N, H, W, Cin, Cout, k, s, p, parity_class=c) for c in range(s * s)]
    assert len(names) == len(kd) and all(_same(a, b) for a, b in zip(names, kd)), names
    dy = _mk(hip, (N, OH, OW, Cout), dt, 46)
    w2 = (_mk(hip, (Cout, k, k, Cin), dt, 47) * (1.0 / (k * k * Cout) ** 0.5)).to(dt)
    wT = ops.weight_transpose(w2)
    xr = torch.zeros((N, Cin, H, W), requires_grad=True)
    F.conv2d(xr, w2.float().cpu().permute(0, 3, 1, 2), stride=s, padding=p).backward(dy.float().cpu().permute(0, 3, 1, 2))
    dref = xr.grad.permute(0, 2, 3, 1)
    dx = ops.conv2d_dgrad(dy, wT, (H, W), s, p)
    tol = 2e-2 * max(1.0, dref.abs().max().item())
    assert (dx.float().cpu() - dref).abs().max().item() <= tol
    if s == 1:      # the Bottleneck shortcut-gradient form: dx = dgrad + residual
        r2 = _mk(hip, (N, H, W, Cin), dt, 48)
        dx2 = ops.conv2d_dgrad(dy, wT, (H, W), s, p, residual=r2)
        assert (dx2.float().cpu() - (dref + r2.float().cpu())).abs().max().item() <= tol + 2e-2 * r2.float().abs().max().item()
    # wgrad
    assert kn("wgrad", dt, N, H, W, Cin, Cout, k, s, p) == kw
    dw = torch.zeros((Cout, k, k, Cin), dtype=torch.float32, device=hip.device)
    ops.conv2d_wgrad(x, dy, dw, k, s, p)
    wr = torch.zeros((Cout, Cin, k, k), requires_grad=True)
    F.conv2d(x.float().cpu().permute(0, 3, 1, 2), wr, stride=s, padding=p).backward(dy.float().cpu().permute(0, 3, 1, 2))
    wref = wr.grad.permute(0, 2, 3, 1)
    assert (dw.cpu() - wref).abs().max().item() <= 1e-4 * max(1.0, wref.abs().max().item())


STRIDE2_WGRAD = [   # (case, kernel): the 64-input-channel down-sampling layers take the row-sharing stride-2 form; the wider ones keep the
    # 256x256 per-tap tile (the 128x128 / four-cin-tile stride-2 forms measured slower, profiles/r04_mb_3x3_wgrad_stride2_ab.txt, and
    # were removed in r05 together with their knob)
    ((2, 24, 24, 64, 128, 3, 2, 1), "conv_wgrad_rs_kernel<128, 64, 2, 2, 2>"),
    ((1, 22, 18, 64, 136, 3, 2, 1), "conv_wgrad_rs_kernel<128, 64, 2, 2, 2>"),           # ragged K-split, 9-pixel output rows, ragged cout tile
    ((2, 20, 20, 128, 256, 3, 2, 1), "conv_wgrad_tr_kernel<256, 256, 2, 4>"),            # >= 128 input channels: per-tap tile
    ((1, 15, 15, 64, 128, 3, 2, 1), "conv_wgrad_tr_kernel<128, 128, 2, 2>"),             # odd input size: not eligible
]


def _wgrad_in_child(hip, case, env, kernel):
    """one weight gradient vs torch in a CHILD process (a fresh library instance per case)"""
    import subprocess, sys, os
    N, H, W, Cin, Cout, k, s_, p_ = case
    code = f"""
import sys, torch
sys.path.insert(0, {repr(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))})
import torch.nn.functional as F
from efficientteacher_amd import _lib, ops
emu = {repr(hip.emulated)}
if emu:
    from tests.simt_emu import build as b
    _lib._use_library_for_tests(b.build(), True)
dev = torch.device("cpu" if emu else "cuda:0")
dt = torch.bfloat16
N, H, W, Cin, Cout, k, s, p = {N}, {H}, {W}, {Cin}, {Cout}, {k}, {s_}, {p_}
name = ops.kernel_name("wgrad", dt, N, H, W, Cin, Cout, k, s, p).replace("<unsigned short, ", "<")
assert name == {repr(kernel)}, name
g = torch.Generator().manual_seed(5)
x = torch.randn(N, H, W, Cin, generator=g).to(dt).to(dev)
OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
dy = torch.randn(N, OH, OW, Cout, generator=g).to(dt).to(dev)
dw = torch.zeros(Cout, k, k, Cin, device=dev)
ops.conv2d_wgrad(x, dy, dw, k, s, p)
wr = torch.zeros(Cout, Cin, k, k, requires_grad=True)
F.conv2d(x.float().cpu().permute(0, 3, 1, 2), wr, stride=s, padding=p).backward(dy.float().cpu().permute(0, 3, 1, 2))
ref = wr.grad.permute(0, 2, 3, 1)
err = (dw.cpu() - ref).abs().max().item()
assert err <= 1e-4 * max(1.0, ref.abs().max().item()), err
print("OK")
"""
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-500:] + r.stderr[-1500:]


@pytest.mark.parametrize("case,kernel", STRIDE2_WGRAD, ids=[str(c) for c, _ in STRIDE2_WGRAD])
def test_wgrad_stride2_row_sharing(hip, case, kernel):
    """conv_wgrad_rs_kernel<..., 2> (two X rows per K-slot) vs torch, and the shapes that must NOT select it"""
    _wgrad_in_child(hip, case, {}, kernel)


def test_wgrad_stride2_row_sharing_grouped(hip):
    """conv_wgrad_rs_kernel<128, 64, 2, 2, 2>: three same-shaped stride-2 3x3 layers in one grouped launch, one of them reading a channel
    slice of a wider buffer; 12-pixel output rows (a 64-slot K chunk spans five padded rows), dY with a pixel stride"""
    from efficientteacher_amd import ops
    N, H, W, Cin, Cout, k = 2, 24, 24, 64, 128, 3
    dt = torch.bfloat16
    assert kn("wgrad", dt, N, H, W, Cin, Cout, k, 2, 1) == "conv_wgrad_rs_kernel<128, 64, 2, 2, 2>"
    items, refs = [], []
    for i in range(3):
        if i == 1:
            wide = _mk(hip, (N, H, W, Cin + 16), dt, 120 + i)
            x = wide[..., 8:8 + Cin]
        else:
            x = _mk(hip, (N, H, W, Cin), dt, 120 + i)
        dyw = _mk(hip, (N, H // 2, W // 2, Cout + 8), dt, 130 + i)
        dy = dyw[..., :Cout] if i == 2 else dyw[..., :Cout].contiguous()
        dw = torch.zeros((Cout, k, k, Cin), dtype=torch.float32, device=hip.device)
        items.append((x, dy, dw))
        wr = torch.zeros((Cout, Cin, k, k), requires_grad=True)
        F.conv2d(x.float().cpu().permute(0, 3, 1, 2), wr, stride=2, padding=1).backward(dy.float().cpu().permute(0, 3, 1, 2))
        refs.append(wr.grad.permute(0, 2, 3, 1))
    ops.conv2d_wgrad_grouped(items, k, 2, 1)
    for (_, _, dw), ref in zip(items, refs):
        assert (dw.cpu() - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("stride,kernel", [(1, "conv_wgrad_rs_kernel<128, 128, 2, 4>"), (2, "conv_wgrad_tr_kernel<256, 256, 2, 4>")])
def test_wgrad_grouped_eight_layers(hip, stride, kernel):
    """the bench's grouped launch: 8 same-shaped 3x3 layers share one K-split -- on the row-sharing kernel (stride 1) and on the
    256x256 per-tap tile (stride 2)"""
    from efficientteacher_amd import ops
    N, H, W, C, k = 1, 10, 10, 256, 3
    dt = torch.bfloat16
    assert kn("wgrad", dt, N, H, W, C, C, k, stride, 1) == kernel
    OH, OW = ops.conv_out_hw(H, W, k, stride, 1)
    items, refs = [], []
    for i in range(8):
        x = _mk(hip, (N, H, W, C), dt, 60 + i)
        dy = _mk(hip, (N, OH, OW, C), dt, 70 + i)
        dw = torch.zeros((C, k, k, C), dtype=torch.float32, device=hip.device)
        items.append((x, dy, dw))
        wr = torch.zeros((C, C, k, k), requires_grad=True)
        F.conv2d(x.float().cpu().permute(0, 3, 1, 2), wr, stride=stride, padding=1).backward(dy.float().cpu().permute(0, 3, 1, 2))
        refs.append(wr.grad.permute(0, 2, 3, 1))
    ops.conv2d_wgrad_grouped(items, k, stride, 1)
    for (_, _, dw), ref in zip(items, refs):
        assert (dw.cpu() - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 12, 12, 64, 40, 3), (2, 20, 20, 256, 256, 3), (3, 10, 10, 32, 128, 1), (2, 9, 11, 256, 256, 1),
                                  (1, 13, 13, 128, 128, 1), (2, 9, 9, 64, 64, 1)],
                         ids=["128x64 tile", "256x256 tile", "1x1", "1x1 stream K=256", "1x1 stream K=128", "1x1 stream K=64"])
def test_dgrad_with_fused_bn_backward_sums(hip, case, dtype):
    """et_conv2d_dgrad_bn + et_bn_act_bwd_from_partials == et_conv2d_dgrad + et_bn_act_bwd (the separate reduce pass),
    with and without the shortcut-gradient residual; and both equal torch autograd of act(BN(y)) on the same tensors."""
    from efficientteacher_amd import ops
    N, H, W, Cin, Cout, k = case
    if hip.emulated and Cin >= 256:      # one ragged 256-row tile pair is enough for the CPU tier (the GPU tier runs the full case)
        N, H, W = 1, 17, 17
        if dtype != torch.bfloat16:
            pytest.skip("fp32 never selects the 256-row tiles; one 16-bit storage type is enough for this size on the emulator (the GPU tier runs both)")
    p = k // 2
    dy = _mk(hip, (N, H, W, Cout), dtype, 81)
    w = (_mk(hip, (Cout, k, k, Cin), dtype, 82) * (1.0 / (k * k * Cout) ** 0.5)).to(dtype)
    wT = ops.weight_transpose(w)
    y = _mk(hip, (N, H, W, Cin), dtype, 83)                      # raw conv output of the PRODUCER block (channels = Cin here)
    res = _mk(hip, (N, H, W, Cin), dtype, 84)
    gamma = torch.rand(Cin, generator=torch.Generator().manual_seed(85)).add(0.5).to(hip.device)
    beta = torch.randn(Cin, generator=torch.Generator().manual_seed(86)).to(hip.device)
    yf = y.float().reshape(-1, Cin)
    mean = yf.mean(0)
    invstd = 1.0 / torch.sqrt(yf.var(0, unbiased=False) + 1e-3)
    scale = gamma * invstd
    shift = beta - mean * scale
    for residual in (None, res):
        for act in (ops.ACT_SILU, ops.ACT_NONE):
            hand = ops.BnBwdSums(y, scale, shift, act)
            dz_f = ops.conv2d_dgrad(dy, wT, (H, W), 1, p, residual=residual, bn=hand)
            assert hand.partial is not None
            dg_f, db_f = torch.zeros(Cin, device=hip.device), torch.zeros(Cin, device=hip.device)
            out_f = ops.bn_act_bwd(dz_f, y, gamma, scale, shift, mean, invstd, act, dg_f, db_f, partial=hand.take(dz_f))
            dz_u = ops.conv2d_dgrad(dy, wT, (H, W), 1, p, residual=residual)
            dg_u, db_u = torch.zeros(Cin, device=hip.device), torch.zeros(Cin, device=hip.device)
            out_u = ops.bn_act_bwd(dz_u, y, gamma, scale, shift, mean, invstd, act, dg_u, db_u)
            assert torch.equal(dz_f, dz_u)                        # the extra epilogue work does not change what is stored
            n = N * H * W
            tol = (2e-4 if dtype == torch.float32 else 2e-2)
            assert torch.allclose(db_f, db_u, rtol=1e-4, atol=tol * n ** 0.5)
            assert torch.allclose(dg_f, dg_u, rtol=1e-3, atol=tol * n ** 0.5)
            assert (out_f.float() - out_u.float()).abs().max().item() <= tol * max(1.0, out_u.float().abs().max().item())
            # torch autograd reference on the stored dz
            yr = y.detach().float().cpu().clone().requires_grad_(True)
            g_r, b_r = gamma.cpu().clone().requires_grad_(True), beta.cpu().clone().requires_grad_(True)
            u = torch.nn.functional.batch_norm(yr.permute(0, 3, 1, 2), None, None, g_r, b_r, True, 0.0, 1e-3)
            z = torch.nn.functional.silu(u) if act == ops.ACT_SILU else u
            z.backward(dz_f.float().cpu().permute(0, 3, 1, 2))
            assert (out_f.float().cpu() - yr.grad).abs().max().item() <= 5 * tol * max(1.0, yr.grad.abs().max().item())
            assert torch.allclose(dg_f.cpu(), g_r.grad, rtol=2e-2, atol=5 * tol * n ** 0.5)
            assert torch.allclose(db_f.cpu(), b_r.grad, rtol=2e-2, atol=5 * tol * n ** 0.5)


@pytest.mark.parametrize("N,H,W,Cout", [(1, 16, 16, 32), (2, 36, 40, 64), (1, 20, 300, 48), (3, 8, 132, 64)])
def test_stem_kernel(hip, N, H, W, Cout):
    """the dedicated 6x6 s2 p2 kernel of the packed image (conv_stem_kernel): ragged tile grids in both directions, channel
    counts below 64, raw output + BN statistics (student) and folded scale/bias + SiLU into a channel slice (teacher)"""
    _stem_case(hip, N, H, W, Cout)


def _stem_case(hip, N, H, W, Cout):
    from efficientteacher_amd import ops
    dt = torch.bfloat16
    assert kn("fwd", dt, N, H, W, 8, Cout, 6, 2, 2) == "conv_stem_kernel"
    x = _mk(hip, (N, H, W, 8), dt, 141)
    x[..., 3:] = 0                                   # the packed image: 3 real channels
    w = (_mk(hip, (Cout, 6, 6, 8), dt, 142) * (1.0 / (36 * 3) ** 0.5)).to(dt)
    OH, OW = ops.conv_out_hw(H, W, 6, 2, 2)
    ref = _ref_conv(x, w, 2, 2)
    y, stats = ops.conv2d_fwd(x, w, 2, 2, want_stats=True)
    tol = 2e-2 * max(1.0, ref.abs().max().item())
    assert (y.float().cpu() - ref).abs().max().item() <= tol
    flat = ref.reshape(-1, Cout)
    st = stats.sum(0).cpu()
    assert torch.allclose(st[0], flat.sum(0), rtol=1e-3, atol=2e-2 * flat.shape[0] ** 0.5 * 4)
    assert torch.allclose(st[1], (flat ** 2).sum(0), rtol=2e-2, atol=1e-3)
    sc = _mk(hip, (Cout,), torch.float32, 143).abs() + 0.5
    bi = _mk(hip, (Cout,), torch.float32, 144)
    wide = torch.zeros((N, OH, OW, Cout + 16), dtype=dt, device=hip.device)
    ops.conv2d_fwd(x, w, 2, 2, scale=sc, bias=bi, act=ops.ACT_SILU, out=wide[..., 8:8 + Cout])
    ref2 = F.silu(ref * sc.cpu() + bi.cpu())
    assert (wide[..., 8:8 + Cout].float().cpu() - ref2).abs().max().item() <= 3e-2 * max(1.0, ref2.abs().max().item())
    assert (wide[..., :8] == 0).all() and (wide[..., 8 + Cout:] == 0).all()
    # the generic kernel on the same problem (reached through the residual form,
    # which the stem kernel declines)
    zero = torch.zeros((N, OH, OW, Cout), dtype=dt, device=hip.device)
    y2, _ = ops.conv2d_fwd(x, w, 2, 2, residual=zero, want_stats=True)
    assert (y2.float() - y.float()).abs().max().item() <= 1e-2 * max(1.0, ref.abs().max().item())


@pytest.mark.gpu
def test_stem_kernel_full_size():
    """the bench's own stem problem (640x640, 64 channels, many persistent tiles per workgroup) against the generic
    gather-GEMM on the same tensors, element-wise, plus the BN statistics"""
    from efficientteacher_amd import _lib, ops
    _lib._use_library_for_tests(None, False)
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    N, H, W, Cout = 16, 640, 640, 64
    g = torch.Generator().manual_seed(7)
    x = torch.zeros((N, H, W, 8), dtype=dt, device=dev)
    x[..., :3] = torch.rand((N, H, W, 3), generator=g).to(dev).to(dt)
    w = (torch.randn((Cout, 6, 6, 8), generator=g) * 0.1).to(dev).to(dt)
    assert kn("fwd", dt, N, H, W, 8, Cout, 6, 2, 2) == "conv_stem_kernel"
    y, st = ops.conv2d_fwd(x, w, 2, 2, want_stats=True)
    zero = torch.zeros_like(y)
    y2, st2 = ops.conv2d_fwd(x, w, 2, 2, residual=zero, want_stats=True)          # the stem kernel declines residuals
    assert (y.float() - y2.float()).abs().max().item() <= 2e-2 * y2.float().abs().max().item()
    a, b = st.double().sum(0), st2.double().sum(0)
    assert torch.allclose(a, b, rtol=1e-4, atol=1e-2)
    ref = y2.float().reshape(-1, Cout)
    assert torch.allclose(a[0], ref.double().sum(0), rtol=2e-3, atol=1.0)


# ---- the benchmarked workloads only launch instantiations that SELECT compares element-wise -------------------------------
def _workload_conv_shapes(wl_name):
    """(H, W, Cin, Cout, k, s, p) of every conv of the workload's model at 640 px, read off the oracle's nn.Conv2d modules
    (forward hooks at 64 px, scaled by 10: the graph is fully convolutional)"""
    import bench
    import torch.nn as nn
    wl = bench.WORKLOADS[wl_name]
    cfg = bench.load_cfg(wl, 64)
    if "v8" in wl_name:
        from oracle import v8 as o
    else:
        from oracle import model as o
    om = o.Model.from_cfg(cfg).eval()
    shapes = []

    def hook(mod, inp, out):
        x = inp[0]
        shapes.append((x.shape[2] * 10, x.shape[3] * 10, mod.in_channels, mod.out_channels, mod.kernel_size[0], mod.stride[0], mod.padding[0]))
    for mm in om.modules():
        if isinstance(mm, nn.Conv2d):
            mm.register_forward_hook(hook)
    with torch.no_grad():
        om(torch.zeros(1, 3, 64, 64))
    return shapes


@pytest.mark.parametrize("wl_name,batches", [("v5l-ssod", (16, 32, 64)), ("v5s-sup", (64,)), ("v8-sup", (32,))])
def test_bench_workloads_launch_only_covered_instantiations(hip_lib_path, wl_name, batches):
    """Every conv launch of every bench.py workload (forward, each dgrad parity class, wgrad; teacher batch and student batch)
    resolves -- through the library's own selection logic, et_conv2d_kernel_name -- to an instantiation that SELECT above
    compares element-wise with torch (pin the YOLOv8 kernels by name as well).  Host logic only."""
    from efficientteacher_amd import _lib, ops
    _lib._use_library_for_tests(None, False)
    # what _check_instantiation runs for a SELECT case: the plain forward, the forward with a residual, the plain dgrad, the dgrad
    # with a residual (test_dgrad_with_fused_bn_backward_sums adds the BN-backward sums on the same instantiation), the wgrad --
    # resolved through the library, and for the tiled kernels equal to the names SELECT spells out
    covered = set()
    for (N, H, W, Cin, Cout, k, s, p), kf, kd, kw in SELECT:
        for op in ("fwd", "fwd_res") + (("dgrad", "dgrad_full", "wgrad") if kd is not None else ()):
            if op == "fwd_res" and k == 6:
                continue
            for pc in (range(s * s) if (op in ("dgrad", "dgrad_full") and s == 2) else (0,)):
                if op == "dgrad_full" and s == 2:
                    continue
                covered.add(kn(op, torch.bfloat16, N, H, W, Cin, Cout, k, s, p, parity_class=pc))
    missing = {}
    for B in batches:
        for (h, w, ci, co, k, s, p) in _workload_conv_shapes(wl_name):
            cip, cop = (8 if k == 6 else (ci + 7) // 8 * 8), (co + 7) // 8 * 8
            for op in (("fwd",) if k == 6 else ("fwd", "fwd_res", "dgrad", "dgrad_full", "wgrad")):
                if op in ("fwd_res", "dgrad_full") and s != 1:
                    continue
                for pc in (range(s * s) if (op == "dgrad" and s == 2) else (0,)):
                    n = kn(op, torch.bfloat16, B, h, w, cip, cop, k, s, p, parity_class=pc)
                    if n not in covered:
                        missing.setdefault(n, (op, B, h, w, ci, co, k, s))
    assert not missing, missing
