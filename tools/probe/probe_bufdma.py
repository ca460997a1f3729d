"""buffer_load_dwordx4 ... lds semantics on gfx950 (see probe_bufdma.hip).  src[i] = i (dwords), LDS pre-filled with 0xdeadbeef."""
import ctypes
import os

import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libprobe_bufdma.so"))
lib.probe_bufdma.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
src = torch.arange(4096, dtype=torch.int32, device="cuda")


def run(name, nrec, voff, soff):
    v = torch.tensor(voff, dtype=torch.int32, device="cuda")
    out = torch.zeros(256, dtype=torch.int32, device="cuda")
    rc = lib.probe_bufdma(src.data_ptr(), nrec, v.data_ptr(), soff, out.data_ptr(), None)
    torch.cuda.synchronize()
    o = out.cpu().view(64, 4)
    print(name, "rc", rc)
    for l in (0, 1, 2, 3, 62, 63):
        print("   lane", l, "voff", voff[l], [hex(int(x) & 0xffffffff) for x in o[l]])


lin = [l * 16 for l in range(64)]
run("in range, soff 0, num_records 1024 B", 1024, lin, 0)
run("lanes 2,3 far out of range (0x7ffffff0)", 1024, [0x7ffffff0 if l in (2, 3) else l * 16 for l in range(64)], 0)
run("num_records 1000: lane 62 straddles (992+16 > 1000), lane 63 beyond", 1000, lin, 0)
run("soff 4096 with num_records 1024: is soffset range-checked?", 1024, lin, 4096)
run("soff 512 with num_records 1024: lanes 32.. beyond if soffset counts", 1024, lin, 512)
run("negative voff (-16) lane 0", 1024, [-16 if l == 0 else l * 16 for l in range(64)], 0)
