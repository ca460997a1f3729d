"""Where does the host spend its time when it enqueues one SSOD step (eager, queue empty at the start)?  cProfile over a few
steps of the package trainer at PER_RANK + PER_RANK images (default 16: configs[3]'s per-rank batch), cumulative + own time."""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

B = int(os.environ.get("PER_RANK", "16"))
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
cfg, tr = bench.build_trainer(dev, -1, 1, 0, B)
rng = np.random.default_rng(1234)
imgs, targets, u_str, u_ori, M_s = bench.make_batch(rng, B, B, cfg.Dataset.img_size, dev)
g = torch.Generator(device="cpu").manual_seed(99)
synth = bench.synth_teacher_scores(cfg, B, cfg.Dataset.img_size, g).to(dev)


def hook(tp):
    tp[..., 4:] = synth
    return tp


tr.teacher_pred_hook = hook
for i in range(4):
    tr.train_instance(imgs, targets, None, u_str, u_ori, None, M_s, 2000 + i)
torch.cuda.synchronize()
rows = []
for i in range(6):
    t0 = time.perf_counter()
    tr.train_instance(imgs, targets, None, u_str, u_ori, None, M_s, 2010 + i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    rows.append(((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
print("HOST enqueue ms (empty queue)", [round(r[0], 2) for r in rows], "step ms", [round(r[1], 2) for r in rows])
pr = cProfile.Profile()
N = 5
for i in range(N):
    pr.enable()
    tr.train_instance(imgs, targets, None, u_str, u_ori, None, M_s, 2020 + i)
    pr.disable()
    torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(f"==== by {key} ({N} steps)")
    print("\n".join(l[:170] for l in s.getvalue().splitlines()[4:60]))
