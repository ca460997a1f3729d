"""Probe: is ONE training step reproducible?  The tiny golden model (tests/test_ssod_step.py::make_trainer), identical inputs and state,
optimizer disabled, fresh trainer per run: the gradient arena of run k against run 0.  Outliers far above the fp32-atomics floor
(~5e-6) mean a race.  Usage: step_determinism.py [runs] [overlap 0|1] [side-stream wgrad 0|1] [bf16|fp32]   (ET_POISON=1, ET_TRACE=1: see below)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.conftest import golden  # noqa: E402
from tests.test_ssod_step import make_trainer  # noqa: E402


class Hip:
    device = torch.device("cuda:0")

    @staticmethod
    def t(a, dtype=None):
        x = torch.as_tensor(np.ascontiguousarray(a))
        return (x.to(dtype) if dtype is not None else x).to("cuda:0")


def poison_uninitialized():
    """ET_POISON=1: every torch.empty / empty_like of a floating dtype comes back filled with NaN, so a kernel that reads memory
    nobody wrote (a partial-statistics row beyond its grid, a padded column) shows up as NaN in the gradients"""
    oe, ol = torch.empty, torch.empty_like

    def empty(*a, **k):
        t = oe(*a, **k)
        return t.fill_(float("nan")) if t.is_floating_point() and t.device.type == "cuda" else t

    def empty_like(*a, **k):
        t = ol(*a, **k)
        return t.fill_(float("nan")) if t.is_floating_point() and t.device.type == "cuda" else t
    torch.empty, torch.empty_like = empty, empty_like


def main():
    if os.environ.get("ET_POISON") == "1":
        poison_uninitialized()
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    overlap = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    side = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    dtype = torch.float32 if (len(sys.argv) > 4 and sys.argv[4] == "fp32") else torch.bfloat16
    from efficientteacher_amd import ops
    ops.WGRAD_QUEUE.use_side = side
    g = golden("ssod_step")
    imgs, u_str, u_ori, M_s, tg = (Hip.t(g[k]) for k in ("imgs", "u_str", "u_ori", "M_s", "targets"))
    trace_on = os.environ.get("ET_TRACE") == "1"
    traces = []
    if trace_on:                      # checksum the output of every backward-side library call, in call order
        cur = []

        def wrap(name):
            f = getattr(ops, name)

            def g_(*a, **kw):
                r = f(*a, **kw)
                outs = r if isinstance(r, (tuple, list)) else (r,)
                for o in outs:
                    if torch.is_tensor(o) and o.is_floating_point():
                        cur.append((name, tuple(o.shape), float(o.double().abs().sum().item())))
                if name == "conv2d_wgrad_grouped":
                    for it_ in a[0]:
                        cur.append((name + ":dw", tuple(it_[2].shape), float(it_[2].double().abs().sum().item())))
                return r
            setattr(ops, name, g_)
        for nm in ("conv2d_dgrad", "bn_act_bwd", "conv2d_wgrad_grouped", "conv2d_fwd", "bn_act_fwd", "bn_finalize", "maxpool5_bwd", "upsample2x_bwd"):
            wrap(nm)
    ref, devs, items0 = None, [], None
    for k in range(runs):
        if trace_on:
            cur.clear()
        cfg, t = make_trainer(Hip, dtype)
        t.overlap_teacher = bool(overlap)
        t.optimizer.step = lambda *a, **kw: None
        t.optimizer.zero_grad = lambda *a, **kw: None
        items = t.train_instance(imgs, tg, None, u_str, u_ori, None, M_s, 500)
        torch.cuda.synchronize()
        gr = t.model.flat_state().grads.clone()
        it = {kk: float(v) for kk, v in items.items() if torch.is_tensor(v) and v.numel() == 1}
        if not torch.isfinite(gr).all():
            bad = []
            for n, p_ in t.model.named_parameters():
                if p_.grad is not None and not torch.isfinite(p_.grad).all():
                    bad.append(n)
            print("run", k, "non-finite gradients in", len(bad), "tensors:", bad[:10], "losses", it)
            break
        if trace_on:
            traces.append(list(cur))
            if k and traces[k] != traces[0]:
                for i_, (x_, y_) in enumerate(zip(traces[0], traces[k])):
                    if x_ != y_:
                        print("run", k, "first differing call #", i_, "of", len(traces[0]), ":", x_, "vs", y_, "| previous call:", traces[0][i_ - 1][:2] if i_ else None)
                        break
        if ref is None:
            ref, items0 = gr, it
        else:
            d = (gr - ref).abs().max().item()
            devs.append(d)
            if d > 1e-3:
                print("run", k, "dev", d, {kk: (it[kk], items0[kk]) for kk in it if abs(it[kk] - items0[kk]) > 1e-6 * max(1.0, abs(items0[kk]))})
        del t
    if devs:
        print(f"overlap={overlap} side={side} runs={runs}: outliers {sum(d > 1e-3 for d in devs)}; max {max(devs):.3e} median {sorted(devs)[len(devs) // 2]:.3e}")


if __name__ == "__main__":
    main()
