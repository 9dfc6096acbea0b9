"""Thin helpers the decomposition path calls (interface of ptdeco.utils.common,
reference src/ptdeco/utils/common.py:7-17: same names, arguments and errors)."""

from __future__ import annotations

import gc
import logging
from typing import Any, TypeVar

import torch

__all__ = [
    "to_device",
    "get_gpu_reserved_memory_gb",
    "free_gpu_reserved_memory",
    "get_num_params",
    "is_compound_module",
    "get_type_name",
    "get_default_device",
    "split_module_parent_child_name",
    "replace_submodule_in_place",
]

logger = logging.getLogger(__name__)

T = TypeVar("T", torch.Tensor, dict[str, torch.Tensor])


def to_device(o: T, device: torch.device) -> T:
    """Tensor or dict of tensors -> device; anything else raises ValueError (common.py:25-36)."""
    if isinstance(o, torch.Tensor):
        return o.to(device)
    if isinstance(o, dict):
        return {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in o.items()}
    raise ValueError(f"Unsupported type {type(o)}")


def get_gpu_reserved_memory_gb() -> float:
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    return sum(torch.cuda.memory_reserved(device=i) for i in range(n)) / 1024.0**3


def free_gpu_reserved_memory() -> None:
    if not torch.cuda.is_available():
        return
    before = get_gpu_reserved_memory_gb()
    gc.collect()
    torch.cuda.empty_cache()
    after = get_gpu_reserved_memory_gb()
    logger.info(f"GPU memory: {before:.2f} -> {after:.2f} GB ({after - before:.2f} GB)")


def get_num_params(m: torch.nn.Module, only_trainable: bool = False) -> int:
    """Parameter count, shared tensors counted once (by data_ptr, common.py:58-63)."""
    seen: dict[int, int] = {}
    for p in m.parameters():
        if only_trainable and not p.requires_grad:
            continue
        seen[p.data_ptr()] = p.numel()
    return sum(seen.values())


def is_compound_module(m: torch.nn.Module) -> bool:
    return next(m.children(), None) is not None


def get_type_name(o: Any) -> str:
    cls = type(o)
    return f"{cls.__module__}.{cls.__name__}"


def get_default_device(module: torch.nn.Module) -> torch.device:
    p = next(module.parameters(), None)
    return torch.device("cpu") if p is None else p.device


def split_module_parent_child_name(target: str) -> tuple[str, str]:
    parent, _, child = target.rpartition(".")
    return parent, child


def replace_submodule_in_place(root_module: torch.nn.Module, submodule_name: str,
                               new_submodule: torch.nn.Module) -> None:
    parent, child = split_module_parent_child_name(submodule_name)
    setattr(root_module.get_submodule(parent), child, new_submodule)
