"""fp32 atomic-add throughput: agent scope vs workgroup scope (XCD-local L2), wgrad-epilogue access pattern."""
import ctypes
import json
import os

import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libprobe_atomic.so"))
lib.probe_atomic.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
res = {}
for ntiles, nblk in ((8, 512), (16, 512), (8, 256), (64, 512)):
    for scope, mode in ((0, 0), (0, 1), (1, 1)):
        dw = torch.zeros(ntiles * 128 * 128, device=dev)
        xcc = torch.full((nblk,), -1, dtype=torch.int32, device=dev)
        for _ in range(2):
            lib.probe_atomic(dw.data_ptr(), ntiles, nblk, scope, mode, xcc.data_ptr(), None)
        torch.cuda.synchronize()
        dw.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.probe_atomic(dw.data_ptr(), ntiles, nblk, scope, mode, xcc.data_ptr(), None)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        total = float(dw.sum().item())
        expect = 10.0 * nblk * 128 * 128
        x = xcc.cpu().tolist()
        rr = all(v == (i % 8) for i, v in enumerate(x))
        key = f"tiles{ntiles}_blk{nblk}_scope{'agent' if scope == 0 else 'wg'}_mode{mode}"
        res[key] = dict(us=round(us, 1), GBps=round(nblk * 65536 / us / 1e3, 1), sum_ok=abs(total - expect) < 1e-3 * expect,
                        total=total, expect=expect, xcc_round_robin=rr, xcc_first16=x[:16])
        print(key, res[key])
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/probe_atomic.json", "w"), indent=1)
