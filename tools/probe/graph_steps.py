"""per-step wall time (synchronised) of the graph-replayed step, with one eager step in the middle (diagnosis)"""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
cfg, tr = bench.build_trainer(dev, -1, 1, 0, 32)
rng = np.random.default_rng(1234)
imgs, targets, u_str, u_ori, M_s = bench.make_batch(rng, 32, 32, 640, dev)
g = torch.Generator(device="cpu").manual_seed(99)
synth = (torch.rand(32, 25200, 81, generator=g) ** torch.cat((torch.full((1,), 16.0), torch.full((80,), 4.0)))).to(dev)
def hook(tp):
    tp[..., 4:] = synth
    return tp
tr.teacher_pred_hook = hook
rows = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    eager = (i == 14)
    tr.use_graph = not eager
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.train_instance(imgs, targets, None, u_str, u_ori, None, M_s, 2000 + i)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    rows.append((i, "eager" if (eager or i < 3) else "graph", round((t1 - t0) * 1e3, 2), round((t2 - t0) * 1e3, 2)))
print("STEPS", rows)
# unsynchronised throughput of 10 replays
tr.use_graph = True
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10):
    tr.train_instance(imgs, targets, None, u_str, u_ori, None, M_s, 3000 + i)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("BURST enqueue ms/step", (t1 - t0) * 100, "total ms/step", (t2 - t0) * 100)
