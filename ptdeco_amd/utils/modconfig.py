"""decompose_config <-> module (interface and JSON format of ptdeco.utils.modconfig,
reference src/ptdeco/utils/modconfig.py:9-14, 21-61, 102-130).

``build_module_from_config`` returns the fused low-rank modules of this package
(``LowRankLinear`` / ``LowRankConv1x1``) when a Sequential config describes a rank-r
pair, so a checkpoint written by ptdeco (or by this package) loads straight onto the
two-GEMM HIP forward; state_dict keys are unchanged ('0.weight', '1.weight', '1.bias').
"""

from __future__ import annotations

import collections
import logging
from typing import Any

import torch

from . import common

__all__ = [
    "get_module_config",
    "build_module_from_config",
    "apply_decompose_config_in_place",
    "MODCONFIG_META_KEY",
]

logger = logging.getLogger(__name__)

MODCONFIG_META_KEY = "__meta__"

_CONV_KEYS = ("in_channels", "out_channels", "kernel_size", "bias", "groups", "padding", "padding_mode", "stride",
              "dilation")


def get_module_config(m: torch.nn.Module) -> dict[str, Any]:
    if isinstance(m, torch.nn.Sequential):
        return {"type": "Sequential", "modules": {k: get_module_config(v) for k, v in m.named_children()}}
    if isinstance(m, torch.nn.Conv2d):
        cfg: dict[str, Any] = {"type": "Conv2d"}
        for k in _CONV_KEYS:
            cfg[k] = (m.bias is not None) if k == "bias" else getattr(m, k)
        return cfg
    if isinstance(m, torch.nn.Linear):
        return {"type": "Linear", "in_features": m.in_features, "out_features": m.out_features,
                "bias": m.bias is not None}
    raise ValueError(f"get_module_config not implemented for {type(m)}")


def _as_tuple(v):
    return tuple(v) if isinstance(v, list) else v


def build_module_from_config(config: dict[str, Any]) -> torch.nn.Module:
    kind = config.get("type")
    if kind == "Linear":
        return torch.nn.Linear(config["in_features"], config["out_features"], bias=config["bias"])
    if kind == "Conv2d":
        return torch.nn.Conv2d(
            in_channels=config["in_channels"], out_channels=config["out_channels"],
            kernel_size=_as_tuple(config["kernel_size"]), groups=config["groups"], bias=config["bias"],
            stride=_as_tuple(config["stride"]), padding=_as_tuple(config["padding"]),
            padding_mode=config["padding_mode"], dilation=_as_tuple(config["dilation"]))
    if kind == "Sequential":
        from ..lowrank import fuse_pair  # local import: lowrank imports this module's sibling ops

        children = collections.OrderedDict((k, build_module_from_config(v)) for k, v in config["modules"].items())
        keys = list(children.keys())
        if keys and keys[0] == "0":
            seq = torch.nn.Sequential(*children.values())
        else:
            seq = torch.nn.Sequential(children)
        return fuse_pair(seq)
    raise ValueError(f"type={kind!r} not supported")


def apply_decompose_config_in_place(module: torch.nn.Module, decompose_config: dict[str, Any]) -> None:
    counter: collections.Counter[str] = collections.Counter()
    for name, cfg in decompose_config.items():
        old = module.get_submodule(name)
        new = build_module_from_config(cfg)
        new.to(common.get_default_device(old))
        common.replace_submodule_in_place(module, name, new)
        counter[common.get_type_name(old)] += 1
    for type_name, count in counter.items():
        logger.info(f"Decomposed {count} instances of {type_name}")
