"""Readers for the committed golden fixtures (tests/golden/*.npz, e2e.json)."""

from __future__ import annotations

import functools
import json
import os

import numpy as np
import torch

import toy_models as tm

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@functools.lru_cache(maxsize=None)
def npz(name: str):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@functools.lru_cache(maxsize=None)
def e2e_meta():
    with open(os.path.join(GOLDEN, "e2e.json")) as f:
        return json.load(f)["scenarios"]


def t(a: np.ndarray, bf16: bool = False) -> torch.Tensor:
    x = torch.from_numpy(np.array(a))
    return x.view(torch.bfloat16) if bf16 else x


def build_model(scn: dict) -> torch.nn.Module:
    z = npz("e2e")
    model = {"MLP3": tm.MLP3, "ConvNet": tm.ConvNet}[scn["arch"]]()
    tm.load_state(model, z, f"model.{scn['model']}.")
    return model


def pool(pid: str, limit=None):
    p = [t(a) for a in npz("e2e")[f"pool.{pid}"]]
    return p[:limit] if limit else p


def targets(mid: str, pid: str, limit=None):
    p = [t(a) for a in npz("e2e")[f"targets.{mid}.{pid}"]]
    return p[:limit] if limit else p


def dwain_streams(scn: dict):
    data = tm.cycle_dicts(pool(scn["pool"]), targets(scn["model"], scn["pool"]))
    lim = scn.get("mpool_len")
    metric = tm.cycle_dicts(pool(scn["mpool"], lim), targets(scn["model"], scn["mpool"], lim))
    return data, metric


def final_state(name: str) -> dict[str, torch.Tensor]:
    z = npz("e2e")
    pre = f"{name}.final."
    return {k[len(pre):]: t(z[k]) for k in z.files if k.startswith(pre)}


def jsonable(cfg):
    """decompose_config after a JSON round trip (tuples become lists)."""
    return json.loads(json.dumps(cfg))


# ---- the bf16 scenario (tests/golden/bf16.npz / bf16.json: gen_golden.py --bf16); bf16 tensors are stored as raw bits
@functools.lru_cache(maxsize=None)
def bf16_meta():
    with open(os.path.join(GOLDEN, "bf16.json")) as f:
        return json.load(f)["scenarios"]


def _bits(a: np.ndarray) -> torch.Tensor:
    x = torch.from_numpy(np.array(a))
    return x.view(torch.bfloat16) if x.dtype == torch.int16 else x


def bf16_model(scn: dict) -> torch.nn.Module:
    z = npz("bf16")
    model = tm.MLP3().bfloat16()
    pre = f"model.{scn['model']}."
    model.load_state_dict({k[len(pre):]: _bits(z[k]) for k in z.files if k.startswith(pre)})
    return model


def bf16_streams(scn: dict):
    z = npz("bf16")
    pools = {pid: [_bits(a) for a in z[f"pool.{pid}"]] for pid in (scn["pool"], scn["mpool"])}
    tg = {pid: [t(a) for a in z[f"targets.{scn['model']}.{pid}"]] for pid in (scn["pool"], scn["mpool"])}
    return (tm.cycle_dicts(pools[scn["pool"]], tg[scn["pool"]]), tm.cycle_dicts(pools[scn["mpool"]], tg[scn["mpool"]]),
            pools[scn["pool"]][0])


def bf16_final(name: str):
    z = npz("bf16")
    pre = f"{name}.final."
    return {k[len(pre):]: _bits(z[k]) for k in z.files if k.startswith(pre)}, _bits(z[f"{name}.final_out"])
