"""How much of a FASTER teacher forward reaches the step?  (VERDICT r05 item 1 asks for stream-K on the teacher's under-filled launches:
at best -22 % on its MFMA-bound launches.)  The teacher runs on a second stream beside the student's forward, so its cost to the step is
not its duration.  This probe shrinks the teacher's work -- it runs the EMA model on the first FRAC of the unlabeled batch and tiles the
result to the full batch (pseudo labels, NMS and the losses keep their full size) -- and reports the step time per FRAC:
    python tools/probe/teacher_sensitivity.py [bench.py arguments]        -> one line per FRAC in {1.0, 0.75, 0.5, 0.25}, alternating"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

B = 32
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
cfg, tr = bench.build_trainer(dev, -1, 1, 0, B)
rng = np.random.default_rng(1234)
imgs, targets, u_str, u_ori, M_s = bench.make_batch(rng, B, B, cfg.Dataset.img_size, dev)
imgs, u_str, u_ori = [(t * 255).round().to(torch.uint8) for t in (imgs, u_str, u_ori)]
synth = bench.synth_teacher_scores(cfg, B, cfg.Dataset.img_size, torch.Generator(device="cpu").manual_seed(99)).to(dev)
tr.teacher_pred_hook = lambda tp: (tp.__setitem__((Ellipsis, slice(4, None)), synth), tp)[1]
ema = tr.ema.ema
orig = ema.forward
FRAC = [1.0]


def shrunk(x, *a, **k):
    n = max(1, int(round(x.shape[0] * FRAC[0])))
    if n == x.shape[0]:
        return orig(x, *a, **k)
    (z, feats), extra = orig(x[:n], *a, **k)
    reps = -(-x.shape[0] // n)
    z = z.repeat(reps, 1, 1)[:x.shape[0]]
    return (z, feats), extra


ema.forward = shrunk


def run(frac, steps=20, warm=4):
    FRAC[0] = frac
    for i in range(warm):
        tr.train_instance(imgs, targets, None, u_str, u_ori, None, M_s, 2000 + i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.train_instance(imgs, targets, None, u_str, u_ori, None, M_s, 2100 + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for overlap in (True, False):
    tr.overlap_teacher = overlap
    for rep in range(2):
        for f in (1.0, 0.75, 0.5, 0.25):
            print(f"overlap {overlap}  teacher on {f:4.2f} of the unlabeled batch: {run(f):6.2f} ms per step", flush=True)
