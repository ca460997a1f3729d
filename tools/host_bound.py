"""Is the SSOD step host-bound?  Per step: host time to enqueue (queue empty at start) vs time to completion."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
cfg, tr = bench.build_trainer(dev, -1, 1, 0, 32)
rng = np.random.default_rng(1234)
imgs, targets, u_str, u_ori, M_s = bench.make_batch(rng, 32, 32, cfg.Dataset.img_size, dev)
g = torch.Generator(device="cpu").manual_seed(99)
pw = torch.cat((torch.full((1,), 16.0), torch.full((80,), 4.0)))
synth = (torch.rand(32, 25200, 81, generator=g) ** pw).to(dev)


def hook(tp):
    tp[..., 4:] = synth
    return tp


tr.teacher_pred_hook = hook
tr.overlap_teacher = os.environ.get("OVERLAP", "1") == "1"
for i in range(3):
    tr.train_instance(imgs, targets, None, u_str, u_ori, None, M_s, 2000 + i)
torch.cuda.synchronize()
rows = []
for i in range(5):
    t0 = time.perf_counter()
    tr.train_instance(imgs, targets, None, u_str, u_ori, None, M_s, 2003 + i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
print("HOST", json.dumps(dict(enqueue_ms=[round(r[0], 2) for r in rows], total_ms=[round(r[1], 2) for r in rows])))
