"""buffer-descriptor LDS-DMA: how far must the next M0 write stay from a buffer_load ... lds?  For each probe library (s_nop N after every
piece) run the failing shapes REPS times and count launches whose output differs from the flat-address form."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
from efficientteacher_amd import ops
dev = torch.device("cuda:0")
bad = tot = 0
for (B, h, cin, cout) in [(64, 160, 64, 64), (8, 160, 128, 64), (4, 320, 64, 64)]:
    g = torch.Generator().manual_seed(1)
    nx, nw = B * h * h * cin, cout * 9 * cin
    pool = torch.empty(nx + nw + 256, dtype=torch.bfloat16, device=dev)      # one allocation: the one-descriptor probe needs both operands < 2^31 bytes apart
    x = pool[:nx].view(B, h, h, cin); x.copy_(torch.randn((B, h, h, cin), generator=g).to(torch.bfloat16))
    w = pool[nx + 128:nx + 128 + nw].view(cout, 3, 3, cin); w.copy_((torch.randn((cout, 3, 3, cin), generator=g) * 0.05).to(torch.bfloat16))
    os.environ["ET_CONV_BUF_DMA"] = "0"
    ref = ops.conv2d_fwd(x, w, 1, 1)
    os.environ["ET_CONV_BUF_DMA"] = "1"
    for rep in range(int(sys.argv[1])):
        y = ops.conv2d_fwd(x, w, 1, 1)
        tot += 1; bad += int(not torch.equal(y, ref))
print("launches", tot, "differing from the flat form", bad)
''' % ROOT
LIBS = os.environ.get("ET_PROBE_LIBS")
for lib in (LIBS.split(",") if LIBS else ["default"] + [f"tools/probe/libet_nops{n}.so" for n in (0, 1, 3, 7, 15)]):
    env = dict(os.environ)
    if lib != "default":
        env["ET_HIP_LIB"] = os.path.join(ROOT, lib)
    r = subprocess.run([sys.executable, "-c", CHILD, sys.argv[1] if len(sys.argv) > 1 else "40"], env=env, capture_output=True, text=True)
    print(f"{lib:34s}", (r.stdout.strip().splitlines() or [r.stderr.strip()[-200:]])[-1], flush=True)
