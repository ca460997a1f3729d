"""Does a just-written tensor come back from the 256 MB Infinity Cache?  For sizes 16..512 MB: (a) write buffer i, then read it back
(producer -> consumer, as conv -> BN pass), (b) the same read after 8 other buffers of that size were written in between (cold).
Read = torch sum over int32 views / copy into a second buffer; times by HIP events.  Usage: python tools/probe/mall_probe.py"""
import torch

dev = torch.device("cuda:0")


def t_ms(fn, n=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


for mb in (16, 32, 64, 105, 160, 210, 420):
    n = mb * (1 << 20) // 2
    bufs = [torch.empty(n, dtype=torch.bfloat16, device=dev) for _ in range(9)]
    dst = torch.empty(n, dtype=torch.bfloat16, device=dev)
    src = bufs[0]
    res = {}
    for mode in ("warm", "cold"):
        def prep():
            src.fill_(1.0)
            if mode == "cold":
                for b in bufs[1:]:
                    b.fill_(2.0)
        times = []
        for _ in range(4):
            prep(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); dst.copy_(src); e1.record(); torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        res[mode] = min(times)
    wr = t_ms(lambda: src.fill_(3.0))
    print(f"{mb:4d} MB  copy after own write {res['warm']*1e3:7.1f} us = {2*mb/1e3/res['warm']*1.048576:5.2f} TB/s | after 8 other writes {res['cold']*1e3:7.1f} us = "
          f"{2*mb/1e3/res['cold']*1.048576:5.2f} TB/s | fill {wr*1e3:7.1f} us = {mb/1e3/wr*1.048576:5.2f} TB/s", flush=True)
    del bufs, dst
