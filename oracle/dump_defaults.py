"""Dump the reference's default config tree (configs/defaults.py:5-322) to YAML.

TEST/BUILD INFRASTRUCTURE, build container only.  The key names and default values are part of the
drop-in boundary (YAMLs under configs/ must load unchanged), so they are taken from the
live reference rather than retyped:  python -m oracle.dump_defaults
"""
import os

import yaml

from . import ref_loader

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "efficientteacher_amd", "configs",
                   "defaults.yaml")


def to_plain(node):
    if isinstance(node, dict):
        return {k: to_plain(v) for k, v in node.items()}
    if isinstance(node, tuple):
        return list(node)
    return node


if __name__ == "__main__":
    cfg = ref_loader.get_cfg()
    with open(OUT, "w") as f:
        f.write("# default config tree of the reference (configs/defaults.py), dumped by oracle/dump_defaults.py\n")
        yaml.safe_dump(to_plain(cfg), f, default_flow_style=None, sort_keys=True)
    print("wrote", OUT)
