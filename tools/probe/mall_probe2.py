"""Which part of a just-written tensor LARGER than the Infinity Cache is still resident?  Write a 210 / 420 MB buffer front to back
(fill), then copy only its FIRST or only its LAST quarter / half: if the cache keeps the most recently written lines, the tail copies
at warm speed (a consumer that walks the tensor back to front would hit), the head at cold speed.  Usage: python tools/probe/mall_probe2.py"""
import torch

dev = torch.device("cuda:0")
for mb in (210, 420):
    n = mb * (1 << 20) // 2
    src = torch.empty(n, dtype=torch.bfloat16, device=dev)
    other = [torch.empty(n, dtype=torch.bfloat16, device=dev) for _ in range(3)]
    for frac in (4, 2):
        m = n // frac
        dst = torch.empty(m, dtype=torch.bfloat16, device=dev)
        out = {}
        for part in ("head", "tail", "cold"):
            ts = []
            for _ in range(4):
                for o in other:
                    o.fill_(2.0)
                src.fill_(1.0)
                if part == "cold":
                    for o in other:
                        o.fill_(2.0)
                torch.cuda.synchronize()
                view = src[:m] if part != "tail" else src[n - m:]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); dst.copy_(view); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            out[part] = min(ts)
        sz = mb / frac
        print(f"{mb} MB written front to back; copy of 1/{frac} ({sz:.0f} MB): head {out['head']*1e3:6.1f} us ({2*sz/1e3/out['head']*1.048576:4.2f} TB/s)  "
              f"tail {out['tail']*1e3:6.1f} us ({2*sz/1e3/out['tail']*1.048576:4.2f} TB/s)  cold {out['cold']*1e3:6.1f} us ({2*sz/1e3/out['cold']*1.048576:4.2f} TB/s)", flush=True)
