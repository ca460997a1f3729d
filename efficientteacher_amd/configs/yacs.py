"""A small YACS-compatible configuration node (the reference vendors facebookresearch/yacs as
configs/yacs.py; this is an independent implementation of the subset its trainers use).

Semantics kept: attribute access to keys, ``merge_from_file`` (YAML), ``merge_from_list``
(``KEY VALUE ...`` with ``literal_eval`` decoding, train.py:34,67), strict type checking on merge
(only tuple<->list is coerced), ``KeyError`` for unknown keys, ``freeze`` / ``defrost`` /
``is_frozen``, ``clone``, ``dump``.
"""
import copy
from ast import literal_eval

import yaml

_VALID = (tuple, list, str, int, float, bool, type(None))


class CfgNode(dict):
    IMMUTABLE = "__immutable__"

    def __init__(self, init_dict=None):
        super().__init__()
        self.__dict__[CfgNode.IMMUTABLE] = False
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    # -- attribute access ---------------------------------------------------------------------------------
    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.is_frozen():
            raise AttributeError(f"Attempted to set {name} to {value}, but CfgNode is immutable")
        if name in self.__dict__:
            raise AttributeError(f"Invalid attempt to modify internal CfgNode state: {name}")
        if not (isinstance(value, _VALID) or isinstance(value, CfgNode)):
            raise AttributeError(f"Invalid type {type(value)} for key {name}")
        self[name] = value

    # -- (de)serialisation ----------------------------------------------------------------------------------
    def _plain(self):
        return {k: (v._plain() if isinstance(v, CfgNode) else (list(v) if isinstance(v, tuple) else v))
                for k, v in self.items()}

    def dump(self, **kwargs):
        return yaml.safe_dump(self._plain(), **kwargs)

    def __str__(self):
        return self.dump(default_flow_style=None)

    __repr__ = dict.__repr__

    # -- merging ---------------------------------------------------------------------------------------------
    @staticmethod
    def _decode(value):
        if isinstance(value, dict) and not isinstance(value, CfgNode):
            return CfgNode(value)
        if not isinstance(value, str):
            return value
        try:
            return literal_eval(value)
        except (ValueError, SyntaxError):
            return value

    @staticmethod
    def _coerce(new, old, full_key):
        if type(new) is type(old):
            return new
        if isinstance(new, tuple) and isinstance(old, list):
            return list(new)
        if isinstance(new, list) and isinstance(old, tuple):
            return tuple(new)
        raise ValueError(f"Type mismatch ({type(old)} vs. {type(new)}) with values ({old} vs. {new}) "
                         f"for config key: {full_key}")

    def _merge(self, other, path):
        for k, v in other.items():
            full = ".".join(path + [k])
            v = self._decode(copy.deepcopy(v))
            if k not in self:
                raise KeyError(f"Non-existent config key: {full}")
            v = self._coerce(v, self[k], full)
            if isinstance(v, CfgNode):
                self[k]._merge(v, path + [k])
            else:
                self[k] = v

    def merge_from_other_cfg(self, other):
        self._assert_mutable()
        self._merge(other, [])

    def merge_from_file(self, cfg_filename):
        with open(cfg_filename, "r") as f:
            self.merge_from_other_cfg(CfgNode(yaml.safe_load(f) or {}))

    def merge_from_list(self, cfg_list):
        self._assert_mutable()
        if len(cfg_list) % 2:
            raise AssertionError(f"Override list has odd length: {cfg_list}; it must be a list of pairs")
        for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
            keys = full_key.split(".")
            d = self
            for sub in keys[:-1]:
                if sub not in d:
                    raise KeyError(f"Non-existent key: {full_key}")
                d = d[sub]
            if keys[-1] not in d:
                raise KeyError(f"Non-existent key: {full_key}")
            d[keys[-1]] = self._coerce(self._decode(v), d[keys[-1]], full_key)

    # -- mutability --------------------------------------------------------------------------------------------
    def _assert_mutable(self):
        if self.is_frozen():
            raise AttributeError("CfgNode is immutable")

    def _set_immutable(self, flag):
        self.__dict__[CfgNode.IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_immutable(flag)

    def freeze(self):
        self._set_immutable(True)

    def defrost(self):
        self._set_immutable(False)

    def is_frozen(self):
        return self.__dict__[CfgNode.IMMUTABLE]

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        new = CfgNode({k: copy.deepcopy(v, memo) for k, v in self.items()})
        new._set_immutable(self.is_frozen())
        return new
