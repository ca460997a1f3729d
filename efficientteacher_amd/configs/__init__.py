from .defaults import get_cfg  # noqa: F401
from .yacs import CfgNode  # noqa: F401
