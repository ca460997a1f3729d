"""fp16-mode loss deviation vs the fp32 oracle (tests/test_step_fullsize.py's 2 + 2 YOLOv5l step) with the BatchNorm statistics on
partial rows + fp64 finalize vs on the sharded fp32 accumulators, a few runs each (the sharded sums depend on atomic order)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from efficientteacher_amd import autograd, ops  # noqa: E402
from tests.test_step_fullsize import run_ssod_step_parity  # noqa: E402

dev = torch.device("cuda:0")
for dtype in (torch.float16, torch.bfloat16):
    for sharded, rows in ((False, 0), (True, 2048), (True, 10 ** 9), (False, 0), (True, 2048)):
        autograd.SHARDED_BN = sharded
        ops.SHARD_MAX_ADDS = rows
        for _ in range(2):
            r = run_ssod_step_parity(dev, dtype, Bl=2, Bu=2, amp_calibration=False)
            print(str(dtype)[6:], "sharded" if sharded else "rows   ", rows, {k: round(v, 5) for k, v in r["loss_rel"].items()}, flush=True)
