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
    return _C.clone()
