"""A/B of the sharded BatchNorm statistics (autograd.SHARDED_BN) on the bench step, same build, same box:
    python tools/probe/bn_sharded_ab.py <max additions per channel, 0 = off> [bench.py arguments]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from efficientteacher_amd import autograd  # noqa: E402

from efficientteacher_amd import ops  # noqa: E402

autograd.SHARDED_BN = sys.argv[1] != "0"          # 0 = partial rows everywhere; N > 0 = sharded where a producer adds <= N times per channel (ops.SHARD_MAX_ADDS)
if autograd.SHARDED_BN:
    ops.SHARD_MAX_ADDS = int(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
