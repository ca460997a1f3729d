"""Run the ds_read_b64_tr_b16 probe with several per-lane address patterns; dump {pattern: [[4 element
indices per lane]]} to gpurun_out/probe_tr.json.  LDS holds u16 value == element index."""
import ctypes
import json
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libprobe_tr.so"))
lib.probe_tr.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
pats = {
    "same0": [0] * 64,
    "linear8": [l * 8 for l in range(64)],
    "rows64B": [l * 64 for l in range(64)],                       # every lane its own 64-byte row
    "grp4x16": [((l % 16) // 4) * 32 + (l % 4) * 8 + (l // 16) * 128 for l in range(64)],   # 4x16 row-major blocks
    "rows256B_cols": [((l % 16) // 4) * 256 + (l % 4) * 8 + ((l // 16) % 2) * 32 + (l // 32) * 2048 for l in range(64)],
}
res = {}
for k, a in pats.items():
    addr = torch.tensor(a, dtype=torch.int32, device="cuda")
    out = torch.zeros(256, dtype=torch.int16, device="cuda")
    rc = lib.probe_tr(addr.data_ptr(), out.data_ptr(), None)
    torch.cuda.synchronize()
    res[k] = dict(rc=rc, addr=a, out=out.cpu().view(64, 4).tolist())
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/probe_tr.json", "w"))
for k in res:
    print(k, res[k]["out"][:20])
