"""Phase timestamps of the LDS-DMA gather-GEMM (experiment build -DET_ABLATE=9, tools/probe/build_ablate.sh 9).
usage: ET_HIP_LIB=tools/probe/libet_abl9.so python tools/probe/ts_conv.py cin cout k s h B"""
import ctypes, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efficientteacher_amd import ops, _lib  # noqa: E402

cin, cout, k, s, h, B = [int(v) for v in sys.argv[1:7]]
dev = torch.device("cuda:0")
x = torch.randn(B, h, h, cin, device=dev).to(torch.bfloat16)
w = (torch.randn(cout, k, k, cin, device=dev) * 0.05).to(torch.bfloat16)
p = k // 2
oh, ow = ops.conv_out_hw(h, h, k, s, p)
y = torch.empty(B, oh, ow, cout, device=dev, dtype=torch.bfloat16)
mode = os.environ.get("TS_MODE", "fwd")
dy = torch.randn(B, oh, ow, cout, device=dev).to(torch.bfloat16)
dw = torch.zeros(cout, k, k, cin, device=dev)
for _ in range(3):
    if mode == "fwd":
        ops.conv2d_fwd(x, w, s, p, out=y, want_stats=True)
    else:
        ops.conv2d_wgrad(x, dy, dw, k, s, p)
torch.cuda.synchronize()
lib = _lib.load()
n = 8 * 16384
buf = (ctypes.c_ulonglong * n)()
lib.et_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.et_debug_read(buf, n)
ts = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
ts = ts[ts[:, 0] > 0]
t0 = ts[:, 0].min()
d = dict(blocks=int(len(ts)), span=int(ts[:, 4].max() - t0),
         prologue=float(np.mean(ts[:, 1] - ts[:, 0])), first_chunk=float(np.mean(ts[:, 2] - ts[:, 1])),
         rest_loop=float(np.mean(ts[:, 3] - ts[:, 2])), epilogue=float(np.mean(ts[:, 4] - ts[:, 3])),
         epi_to_last_stage_write=float(np.mean(ts[:, 5] - ts[:, 3])), epi_to_passes_done=float(np.mean(ts[:, 6] - ts[:, 3])),
         block_total=float(np.mean(ts[:, 4] - ts[:, 0])),
         start_spread=[int(v) for v in np.percentile(ts[:, 0] - t0, [0, 25, 50, 75, 100])])
print("TS", json.dumps(dict(shape=[cin, cout, k, s, h, B], rc=rc, **d)))
