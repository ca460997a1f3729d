"""Probe: the EMA teacher's forward (eval mode, BN folded into the conv epilogues) ALONE on the GPU -- B unlabeled images, bf16 --
against the student's train-mode forward at the same batch.  In the step the two overlap on two streams, so their per-kernel
durations in the launch table are inflated by each other; this is the unshared cost.
    python tools/probe/teacher_fwd.py [B]        (under rocprofv3 --kernel-trace --stats for the per-kernel split)
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def timed(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda:0")
    cfg, tr = bench.build_trainer(dev, 0, 1, 0, B)
    rng = np.random.default_rng(0)
    imgs, targets, u_str, u_ori, M_s = bench.make_batch(rng, B, B, 640, dev)
    teacher, student = tr.ema.ema, tr.model
    out = {"B": B}
    with torch.no_grad():
        out["teacher_eval_fwd_ms"] = timed(lambda: teacher(u_ori, augment=False))
        student.train()
        out["student_train_fwd_ms_no_grad"] = timed(lambda: student(u_ori))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
