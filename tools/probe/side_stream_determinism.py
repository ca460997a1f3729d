"""Is the weight-gradient side stream (ops.WGRAD_QUEUE.use_side, ET_WGRAD_STREAM=1) race-free?  N main-stream and N side-stream runs of
the tiny-model SSOD step at the conditioned init (BatchNorm weights 0.3: run-to-run differences stay at rounding level), alternating;
prints every run's distance to the first main-stream run and, for the runs that deviate, WHICH parameter gradients differ.
    python tools/probe/side_stream_determinism.py [N]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.conftest import golden          # noqa: E402
from tests.test_ssod_step import make_trainer   # noqa: E402


class Hip:
    device = torch.device("cuda:0")

    @staticmethod
    def t(a, dtype=None):
        x = torch.as_tensor(np.ascontiguousarray(a))
        return (x.to(dtype) if dtype is not None else x).to("cuda:0")


def main():
    from efficientteacher_amd import ops
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    g = golden("ssod_step")
    imgs, u_str, u_ori, M_s, tg = (Hip.t(g[k]) for k in ("imgs", "u_str", "u_ori", "M_s", "targets"))
    runs = []
    for i in range(2 * n):
        side = bool(i & 1)
        ops.WGRAD_QUEUE.use_side = side
        cfg, t = make_trainer(Hip, torch.bfloat16, bn_gamma=0.3)
        t.optimizer.step = lambda *a, **k: None
        t.optimizer.zero_grad = lambda *a, **k: None
        items = t.train_instance(imgs, tg, None, u_str, u_ori, None, M_s, 500)
        torch.cuda.synchronize()
        fs = t.model.flat_state()
        grads = fs.grads.clone()
        names = {name: (p.grad if p.grad is not None else None) for name, p in t.model.named_parameters()}
        runs.append((side, grads, {k: float(v) for k, v in items.items()}, t))
    ref = runs[0][1]
    scale = ref.abs().max().item()
    print(f"gradient scale {scale:.4e}")
    for i, (side, gr, items, t) in enumerate(runs):
        d = (gr - ref).abs()
        print(f"run {i:2d} {'side' if side else 'main'}  max|g - g_main0| = {d.max().item():.3e}   #elements > 1e-4 scale: {(d > 1e-4 * scale).sum().item()}"
              f"   box {items['box']:.6f} ss_obj {items['ss_obj']:.6f}")
        if d.max().item() > 1e-4 * scale:
            fs = t.model.flat_state()
            base = fs.grads.data_ptr()
            worst = []
            for s in fs.conv_slots.values():
                o = (s.gw.data_ptr() - base) // 4
                dd = d[o:o + s.gw.numel()].max().item()
                if dd > 1e-4 * scale:
                    worst.append((dd, s.index, tuple(s.gw.shape)))
            worst.sort(reverse=True)
            print("     conv weight gradients that differ:", [(f"{a:.2e}", b, c) for a, b, c in worst[:12]], f"... {len(worst)} layers")


if __name__ == "__main__":
    main()
