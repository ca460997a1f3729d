"""what adding the BN totals from every conv tile (same 2*C addresses, fp64 agent-scope atomics) would cost; see the .hip"""
import ctypes, json, os
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libprobe_stat_atomics.so"))
lib.probe_stat_atomics.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
res = {}
for C in (256, 512):
    for nblk, spin in ((400, 7_000_000 // 100), (1600, 1_300_000 // 100), (3200, 1_300_000 // 100), (256, 0)):
        totals = torch.zeros(2 * C, dtype=torch.float64, device=dev); ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        out = torch.zeros(C, device=dev)
        row = {}
        for mode in (0, 1, 2):
            for _ in range(3):
                lib.probe_stat_atomics(totals.data_ptr(), ticket.data_ptr(), out.data_ptr(), C, nblk, spin, mode, None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                lib.probe_stat_atomics(totals.data_ptr(), ticket.data_ptr(), out.data_ptr(), C, nblk, spin, mode, None)
            e1.record(); torch.cuda.synchronize()
            row[f"mode{mode}_us"] = round(e0.elapsed_time(e1) / 20 * 1e3, 2)
        res[f"C{C}_blk{nblk}_spin{spin}"] = row
        print(f"C{C}_blk{nblk}_spin{spin}", row)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/probe_stat_atomics.json", "w"), indent=1)
