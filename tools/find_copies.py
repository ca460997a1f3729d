"""Which big torch-side copies / elementwise ops does one SSOD step still issue?  (torch.profiler, grouped by the
Python source line that launched them)."""
import os
import sys

import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
cfg, tr = bench.build_trainer(dev, -1, 1, 0, 32)
rng = np.random.default_rng(1234)
imgs, targets, u_str, u_ori, M_s = bench.make_batch(rng, 32, 32, cfg.Dataset.img_size, dev)
for i in range(2):
    tr.train_instance(imgs, targets, None, u_str, u_ori, None, M_s, 2000 + i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.train_instance(imgs, targets, None, u_str, u_ori, None, M_s, 2003)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6):
    if e.key.startswith("aten::") and e.device_time_total > 30:
        st = [s for s in e.stack if "efficientteacher_amd" in s or "bench.py" in s]
        rows.append((e.device_time_total, e.count, e.key, str(e.input_shapes)[:80], st[:2]))
rows.sort(reverse=True)
for r in rows[:40]:
    print("%8.0f us  x%-3d %-28s %s\n            %s" % r)
# the many SMALL ones: every aten / memcpy entry by call count
print("---- by count")
small = []
for e in prof.key_averages(group_by_stack_n=8):
    if e.count >= 8 and (e.key.startswith("aten::") or "emcpy" in e.key or "emset" in e.key):
        st = [s for s in e.stack if "efficientteacher_amd" in s or "bench.py" in s]
        small.append((e.count, e.device_time_total, e.key, st[:3]))
small.sort(reverse=True)
for r in small[:40]:
    print("x%-4d %8.0f us  %-30s %s" % r)
