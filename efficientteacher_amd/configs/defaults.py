"""``get_cfg()`` -- a fresh copy of the default config tree (reference configs/defaults.py:325).

The tree itself (every key of reference defaults.py:5-322) lives in ``defaults.yaml`` next to this
file; it is produced from the live reference by ``oracle/dump_defaults.py``.
"""
import os

import yaml

from .yacs import CfgNode

_HERE = os.path.dirname(os.path.abspath(__file__))
_C = None


def get_cfg():
    global _C
    if _C is None:
        with open(os.path.join(_HERE, "defaults.yaml")) as f:
            _C = CfgNode(yaml.safe_load(f))
        # EXTENSION keys of this package (not in the reference's tree; reference yaml files merge unchanged):
        #   Model.deterministic_bn  True = bit-reproducible BatchNorm statistics in the bf16 / fp16 compute modes (partial rows + fp64
        #                           finalize, the fp32 parity mode's path) instead of the sharded fp32 atomics (FlatState(deterministic=))
        _C.Model.deterministic_bn = False
    return _C.clone()
