"""A/B of autograd.FUSE_BN_BWD_K (which dgrads carry their producer's BatchNorm-backward sums: bit 1 = 1x1 layers, bit 2 = k > 1
layers, bit 4 = k > 1 layers on the 128-row row-shift tiles) on the bench step, same build, same box:
    python tools/probe/fuse_bn_bwd_ab.py <K> [bench.py arguments]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from efficientteacher_amd import autograd  # noqa: E402

autograd.FUSE_BN_BWD_K = int(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
