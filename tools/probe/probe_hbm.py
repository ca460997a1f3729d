"""COLD streaming rates of the MI355X for the access forms the step uses (tools/probe/probe_hbm.hip): buffers of 2 GB each (the
Infinity Cache is 256 MB), HIP events, TB/s of bytes actually moved.  Also a WARM variant (64 MB buffers, just written)."""
import ctypes
import json
import os
import subprocess

import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libprobe_hbm.so")
lib = ctypes.CDLL(so)
lib.probe_hbm.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int,
                          ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
out = torch.zeros(1 << 16, device=dev)
res = {}
for label, nbytes in (("cold_2GB", 2 << 30), ("warm_64MB", 64 << 20)):
    a = torch.empty(nbytes // 4, dtype=torch.int32, device=dev).random_(0, 1000)
    b = torch.empty(nbytes // 4, dtype=torch.int32, device=dev).random_(0, 1000)
    c = torch.empty_like(a)
    n_vec = nbytes // 16
    for blocks in (1024, 2048, 4096):
        for mode, unroll, name, moved in ((0, 1, "load x1", 1), (0, 2, "load x2", 1), (0, 4, "load x4", 1), (0, 8, "load x8", 1), (1, 4, "nt load x4", 1),
                                          (2, 0, "lds-dma", 1), (4, 0, "two streams x2", 2), (3, 0, "copy x2", 2)):
            st = torch.cuda.current_stream().cuda_stream
            reps = 3 if nbytes > (1 << 30) else 20
            if label.startswith("warm"):
                a.add_(1); b.add_(1)                    # just written: resident in the Infinity Cache
            lib.probe_hbm(mode, unroll, a.data_ptr(), b.data_ptr(), c.data_ptr(), n_vec, blocks, out.data_ptr(), st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                lib.probe_hbm(mode, unroll, a.data_ptr(), b.data_ptr(), c.data_ptr(), n_vec, blocks, out.data_ptr(), st)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            tbps = moved * nbytes / ms / 1e9
            res[f"{label} blocks{blocks} {name}"] = round(tbps, 2)
            print(f"{label:10s} blocks {blocks:5d}  {name:16s} {ms * 1e3:9.1f} us  {tbps:5.2f} TB/s", flush=True)
    del a, b, c
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/probe_hbm.json", "w"), indent=1)
