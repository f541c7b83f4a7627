"""YAML -> flat attribute dict, the configuration surface of the DIM hot path.

Mirrors the behaviour of the reference loader (reference
``code/base/config.py:10-88``): every top-level YAML section is flattened into one
namespace (``cfg.hidden_size`` not ``cfg.NETWORK.hidden_size``), values are reachable
as attributes and as items, and ``merge_cfg_from_list`` applies ``KEY VALUE`` pairs
with literal decoding and type coercion against the existing value.
"""
import copy
import os
from ast import literal_eval

import yaml

DEFAULT_CONFIG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config.yaml")


class CfgNode(dict):
    """dict with attribute access; nested dicts become CfgNodes."""

    def __init__(self, init_dict=None, key_list=None, new_allowed=False):
        init_dict = {} if init_dict is None else dict(init_dict)
        key_list = [] if key_list is None else key_list
        for k, v in list(init_dict.items()):
            if type(v) is dict:
                init_dict[k] = CfgNode(v, key_list=key_list + [k])
        super().__init__(init_dict)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __repr__(self):
        return "{}({})".format(self.__class__.__name__, super().__repr__())

    def __str__(self):
        lines = []
        for k, v in sorted(self.items()):
            if isinstance(v, CfgNode):
                body = "\n".join("  " + ln for ln in str(v).split("\n"))
                lines.append("{}:\n{}".format(k, body))
            else:
                lines.append("{}: {}".format(k, v))
        return "\n".join(lines)


def load_cfg_from_cfg_file(file):
    """Flatten all sections of ``file`` (must end in .yaml) into one CfgNode."""
    assert os.path.isfile(file) and file.endswith(".yaml"), "{} is not a yaml file".format(file)
    with open(file, "r") as f:
        raw = yaml.safe_load(f)
    flat = {}
    for section in raw:
        for k, v in raw[section].items():
            flat[k] = v
    return CfgNode(flat)


def _decode(v):
    if not isinstance(v, str):
        return v
    try:
        return literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def _coerce(new, old, key, full_key):
    tn, to = type(new), type(old)
    if tn is to or old is None:
        return new
    for a, b in ((tuple, list), (list, tuple)):
        if tn is a and to is b:
            return b(new)
    if to is float and tn is int:
        return float(new)
    raise ValueError("Type mismatch ({} vs. {}) for config key: {}".format(to, tn, full_key))


def merge_cfg_from_list(cfg, cfg_list):
    """Apply ``[KEY, VALUE, KEY, VALUE, ...]`` overrides; only the last dotted
    component of KEY is used (the namespace is flat)."""
    new_cfg = copy.deepcopy(cfg)
    assert len(cfg_list) % 2 == 0
    for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
        sub = full_key.split(".")[-1]
        assert sub in cfg, "Non-existent key: {}".format(full_key)
        setattr(new_cfg, sub, _coerce(_decode(v), cfg[sub], sub, full_key))
    return new_cfg
