"""Does one eager SSOD step synchronise the host with the device?  torch's sync-debug mode reports every blocking call that
goes through torch (copies to the host, .item(), nonzero ...)."""
import os, sys, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402  (reuses its trainer / batch construction)

if __name__ == "__main__":
    sys.argv = ["bench.py", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"]
    orig = torch.cuda.synchronize
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        torch.cuda.set_sync_debug_mode("warn")
        try:
            bench.main()
        finally:
            torch.cuda.set_sync_debug_mode("default")
    msgs = {}
    for x in w:
        if "synchroniz" in str(x.message).lower():
            key = f"{x.filename.split('/')[-1]}:{x.lineno}"
            msgs[key] = msgs.get(key, 0) + 1
    print("SYNC SITES", msgs)
