"""CPU oracle for the ptdeco covariance / eigenvector / rank-search hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``ptdeco_amd/`` may import this
module; it is the checker used by ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py``.

It is a restatement (own code, plain torch-CPU ops) of the reference algorithm
in TCLResearchEurope/ptdeco v0.5.9.  Every function cites the reference
``file:line`` it follows (paths relative to the reference checkout):

* ``dwain.py`` = ``src/ptdeco/dwain/decomposition.py``
* ``falor.py`` = ``src/ptdeco/falor/decomposition.py``
* ``losses.py`` = ``src/ptdeco/utils/losses_primitives.py``
* ``modconfig.py`` = ``src/ptdeco/utils/modconfig.py``

The arithmetic of the path lives in PyTorch (``einsum``, ``matmul``,
``linalg.eigh``); the oracle calls the same ATen ops on CPU so that, run in the
same container, it reproduces the reference bit for bit.  It is pinned by
``tests/golden/*.npz`` which were produced by importing the reference itself
(``tests/golden/gen_golden.py``); see ``tests/test_oracle_golden.py``.

Design differences from the reference (results identical): the layer input is
captured with a forward-pre hook instead of a wrapper module, and every
per-candidate decision is recorded in a ``trace`` list so tests can compare
rank decisions one by one.
"""

from __future__ import annotations

from typing import Any, Optional

import torch

DAMP_FACTOR = 0.01  # dwain.py:14, falor.py:22 (EIGEN_DAMPEN_FACTOR)
META_KEY = "__meta__"  # modconfig.py:18


# --------------------------------------------------------------------------
# metric primitives
# --------------------------------------------------------------------------
def nsr(x: torch.Tensor, y: torch.Tensor, non_channel_dim=(0, 2, 3), eps: float = 1e-3) -> torch.Tensor:
    """Per-channel noise-to-signal ratio, averaged over channels (losses.py:10-22).

    ``torch.std`` is the unbiased estimator; the reference's ``mode`` argument
    is ignored there and therefore absent here.
    """
    var_y = torch.std(y, dim=non_channel_dim) ** 2
    mse = ((x - y) ** 2).mean(dim=non_channel_dim)
    return (mse / (var_y + eps)).mean()


def kl_div(q_logits: torch.Tensor, p_logits: torch.Tensor) -> torch.Tensor:
    """KL(p || q) over softmax(dim=-1), summed over dim 1 (losses.py:48-54)."""
    q = torch.softmax(q_logits, dim=-1)
    p = torch.softmax(p_logits, dim=-1)
    return (p * torch.log(p / q)).sum(dim=1)


def kl_loss(student: torch.Tensor, teacher: torch.Tensor) -> torch.Tensor:
    """mean(max(KL(t||s), KL(s||t))) (losses.py:57-63)."""
    return torch.maximum(kl_div(student, teacher), kl_div(teacher, student)).mean()


# --------------------------------------------------------------------------
# covariance accumulation and eigenvectors
# --------------------------------------------------------------------------
def update_eyyt(eyyt: torch.Tensor, y: torch.Tensor) -> None:
    """``Eyyt += y^T y / T`` with the product formed in y's dtype (dwain.py:147-152)."""
    eyyt += torch.einsum("bp,bq->pq", y, y) / y.shape[0]


def damped_eigvecs(e: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """Tikhonov damping then ``eigh`` (dwain.py:155-163).  Mutates ``e`` as the
    reference does.  Returns (eigenvalues ascending, eigenvectors in columns)."""
    damp = DAMP_FACTOR * torch.mean(torch.diag(e))
    idx = torch.arange(e.shape[-1])
    e[idx, idx] = e[idx, idx] + damp
    w, u = torch.linalg.eigh(e)
    return w, u


def dwain_eigvecs_from_batches(weight: torch.Tensor, xs: list[torch.Tensor], float64: bool = True):
    """dwain.py:211-244 given the captured layer inputs ``xs`` ([T, n_in] each)."""
    n = weight.shape[0]
    eyyt = torch.zeros((n, n), dtype=torch.float64 if float64 else torch.float32)
    for x in xs:
        update_eyyt(eyyt, x @ weight.T)
    eyyt_mean = eyyt / len(xs)
    w, u = damped_eigvecs(eyyt_mean)
    return eyyt, w, u


def falor_accumulate(ey: torch.Tensor, eyyt: torch.Tensor, weight: torch.Tensor, x: torch.Tensor) -> None:
    """falor.py:156-162."""
    y = x @ weight.T
    eyyt += torch.einsum("bp,bq->pq", y, y) / y.shape[0]
    ey += y.mean(dim=0)


def falor_finalize(ey: torch.Tensor, eyyt: torch.Tensor, steps: int, use_mean: bool, use_damping: bool):
    """falor.py:192-208 including quirk: damping is written into ``Eyyt`` after
    ``cov`` was formed, so it only takes effect when ``use_mean`` is False."""
    ey /= steps
    eyyt /= steps
    cov = eyyt - torch.outer(ey, ey) if use_mean else eyyt
    if use_damping:
        damp = DAMP_FACTOR * torch.mean(torch.diag(cov))
        idx = torch.arange(cov.shape[-1])
        eyyt[idx, idx] += damp
    w, u = torch.linalg.eigh(cov)
    return w, u


def falor_eigvecs_from_batches(weight, xs, use_float64=True, use_mean=False, use_damping=True):
    """falor.py:165-208 given the captured layer inputs."""
    n = weight.shape[0]
    dt = torch.float64 if use_float64 else torch.float32
    ey = torch.zeros(n, dtype=dt)
    eyyt = torch.zeros((n, n), dtype=dt)
    for x in xs:
        falor_accumulate(ey, eyyt, weight, x)
    w, u = falor_finalize(ey, eyyt, len(xs), use_mean, use_damping)
    return ey, eyyt, w, u


# --------------------------------------------------------------------------
# factor construction
# --------------------------------------------------------------------------
def factors(weight: torch.Tensor, u: torch.Tensor, rank: int, dtype: torch.dtype):
    """Top-``rank`` eigenvectors -> (U [n_in,r], V [r,n_out], W_deco [n_out,n_in]).

    dwain.py:424-429 (dtype = weight dtype), falor.py:346-348 (dtype = float32).
    """
    uk = u[:, u.shape[1] - rank:].to(dtype)
    big_u = weight.T @ uk
    big_v = uk.T
    return big_u, big_v, (big_u @ big_v).T


def canonical_sign(u: torch.Tensor) -> torch.Tensor:
    """Flip each column so its largest-magnitude entry is positive (test helper)."""
    idx = u.abs().argmax(dim=0)
    s = torch.sign(u[idx, torch.arange(u.shape[1])])
    s[s == 0] = 1
    return u * s


# --------------------------------------------------------------------------
# integer bookkeeping
# --------------------------------------------------------------------------
def params_for_proportion(proportion: float, n_in: int, n_out: int) -> int:
    """dwain.py:319-330."""
    baseline = n_in * n_out
    proposed = (n_in + n_out) * proportion * min(n_in, n_out)
    return int(proposed) if proposed < baseline else baseline


def is_num_params_reduced(proportion: float, n_in: int, n_out: int) -> bool:
    """dwain.py:569-577, falor.py:273-281."""
    return (n_in + n_out) * proportion * min(n_in, n_out) < n_in * n_out


def dwain_candidate_ranks(full_rank: int, min_rank: int, reduction_factor: float) -> list[int]:
    """dwain.py:407-408 -- the last candidate may fall below ``min_rank``."""
    out, r = [], full_rank
    while r > min_rank:
        r = int(r * reduction_factor)
        out.append(r)
    return out


def is_decomposeable(m: torch.nn.Module) -> bool:
    """dwain.py:540-546, falor.py:402-408."""
    return isinstance(m, torch.nn.Linear) or (
        isinstance(m, torch.nn.Conv2d) and tuple(m.kernel_size) == (1, 1) and m.groups == 1
    )


def split_chunks(names: list[str], num_splits: int) -> list[list[str]]:
    """dwain.py:648-660 chunking of the precompute pass."""
    chunk = len(names) // num_splits
    if chunk == 0:
        chunk, num_splits = 1, len(names)
    parts = num_splits if len(names) % num_splits == 0 else num_splits + 1
    return [names[i * chunk:(i + 1) * chunk] for i in range(parts)]


# --------------------------------------------------------------------------
# module plumbing (hook based; the reference uses wrapper modules
# dwain.py:41-144 / falor.py:51-153 with identical observable behaviour)
# --------------------------------------------------------------------------
class _Tap:
    """Captures the last input of a Linear / 1x1 Conv2d as a [T, n_in] matrix."""

    def __init__(self, layer: torch.nn.Module):
        self.layer = layer
        self.is_conv = isinstance(layer, torch.nn.Conv2d)
        self.last: Optional[torch.Tensor] = None
        self.handle = layer.register_forward_pre_hook(self._hook)

    def _hook(self, _mod, args):
        self.last = args[0]

    def rows(self) -> torch.Tensor:
        x = self.last
        if self.is_conv:  # dwain.py:116
            return x.permute(0, 2, 3, 1).reshape(-1, self.layer.in_channels)
        return x.reshape(-1, self.layer.in_features)  # dwain.py:64

    def weight2d(self) -> torch.Tensor:
        w = self.layer.weight.detach()
        return (w[..., 0, 0] if self.is_conv else w).clone()  # dwain.py:58,110

    def set_weight(self, w2d: torch.Tensor) -> None:
        if self.is_conv:
            self.layer.weight.copy_(w2d[:, :, None, None])  # dwain.py:113
        else:
            self.layer.weight.copy_(w2d)  # dwain.py:61

    def close(self) -> None:
        self.handle.remove()


def build_pair(layer: torch.nn.Module, a: torch.Tensor, b: torch.Tensor) -> torch.nn.Sequential:
    """Rank-r pair (dwain.py:69-85, 121-144; falor.py:79-95, 131-153).

    ``a`` = U^T [r, n_in] becomes the first weight, ``b`` = V^T [n_out, r] the
    second; a 1x1 conv pair uses default stride/padding/dilation (quirk 5).
    """
    r = a.shape[0]
    has_bias = layer.bias is not None
    if isinstance(layer, torch.nn.Conv2d):
        m1 = torch.nn.Conv2d(layer.in_channels, r, kernel_size=1, bias=False)
        m2 = torch.nn.Conv2d(r, layer.out_channels, kernel_size=1, bias=has_bias)
        with torch.no_grad():
            m1.weight.copy_(a[:, :, None, None])
            m2.weight.copy_(b[:, :, None, None])
    else:
        m1 = torch.nn.Linear(layer.in_features, r, bias=False)
        m2 = torch.nn.Linear(r, layer.out_features, bias=has_bias)
        m1.weight.data = a[:, :]
        m2.weight.data = b[:, :]
    if has_bias:
        with torch.no_grad():
            m2.bias.copy_(layer.bias)
    return torch.nn.Sequential(m1, m2)


def _set_submodule(root: torch.nn.Module, name: str, new: torch.nn.Module) -> None:
    parent, _, child = name.rpartition(".")
    setattr(root.get_submodule(parent), child, new)


def module_config(m: torch.nn.Module) -> dict[str, Any]:
    """modconfig.py:21-61."""
    if isinstance(m, torch.nn.Sequential):
        return {"type": "Sequential", "modules": {k: module_config(v) for k, v in m.named_children()}}
    if isinstance(m, torch.nn.Conv2d):
        return {
            "type": "Conv2d", "in_channels": m.in_channels, "out_channels": m.out_channels,
            "kernel_size": m.kernel_size, "bias": m.bias is not None, "groups": m.groups,
            "padding": m.padding, "padding_mode": m.padding_mode, "stride": m.stride,
            "dilation": m.dilation,
        }
    if isinstance(m, torch.nn.Linear):
        return {"type": "Linear", "in_features": m.in_features, "out_features": m.out_features,
                "bias": m.bias is not None}
    raise ValueError(f"module_config not implemented for {type(m)}")


def num_params(m: torch.nn.Module) -> int:
    """common.py:58-63 (unique by data_ptr)."""
    return sum(p.numel() for p in {p.data_ptr(): p for p in m.parameters()}.values())


# --------------------------------------------------------------------------
# dwain end to end
# --------------------------------------------------------------------------
class _CovLinear(torch.nn.Module):
    """Stand-in layer of the all-layers precompute pass (dwain.py:166-208)."""

    def __init__(self, weight, bias, float64: bool):
        super().__init__()
        self.weight, self.bias = weight, bias
        n = weight.shape[0]
        self.eyyt = torch.zeros((n, n), dtype=torch.float64 if float64 else torch.float32)
        self.steps = 0

    def forward(self, x):
        y = x @ self.weight.T
        update_eyyt(self.eyyt, y.reshape(-1, self.weight.shape[0]))
        if self.bias is not None:
            y += self.bias
        self.steps += 1
        return y

    def eigvecs(self):
        _, u = damped_eigvecs(self.eyyt / self.steps)
        return u.to(self.weight.dtype)


def dwain_precompute(module, names, num_splits, num_data_steps, data_iterator, float64):
    """dwain.py:580-674."""
    out: dict[str, torch.Tensor] = {}
    for part in split_chunks(names, num_splits):
        originals = {}
        for name in part:
            old = module.get_submodule(name)
            originals[name] = old
            _set_submodule(module, name, _CovLinear(old.weight, old.bias, float64))
        module.eval()
        with torch.no_grad():
            for _ in range(num_data_steps):
                module(next(data_iterator))
        for name in part:
            out[name] = module.get_submodule(name).eigvecs()
        for name in part:
            _set_submodule(module, name, originals[name])
    assert len(out) == len(names)
    return out


def dwain_process_layer(*, module, name, data_iterator, loss_fn, metric_iterator, num_data_steps,
                        num_metric_steps, total_params, nsr_final_threshold, min_rank, trade_off_factor,
                        reduction_factor, max_accepted_ppl_diff, float64, u=None, trace=None):
    """dwain.py:333-537."""
    layer = module.get_submodule(name)
    dtype = layer.weight.dtype
    tap = _Tap(layer)
    try:
        w0 = tap.weight2d()
        n_out, n_in = w0.shape
        full = min(n_in, n_out)
        if full == 1:
            return {"proportion": 1.0, "nsr_final": 0.0, "ppl_final": 0.0, "decomposed_module": None}
        module.eval()
        if u is None:
            eyyt = torch.zeros((n_out, n_out), dtype=torch.float64 if float64 else torch.float32)
            for _ in range(num_data_steps):
                module(next(data_iterator))
                update_eyyt(eyyt, tap.rows() @ w0.T)
            _, u = damped_eigvecs(eyyt / num_data_steps)

        tried = False
        rank_best, nsr_best, ppl_best, drop = full, 0.0, 0.0, 0
        i = 1
        for rank_new in dwain_candidate_ranks(full, min_rank, reduction_factor):
            drop = params_for_proportion(1.0, n_in, n_out) - params_for_proportion(rank_new / full, n_in, n_out)
            frac = drop / total_params
            thr = frac * trade_off_factor
            if drop == 0:  # dwain.py:418-421
                continue
            _, _, w_deco = factors(w0, u, rank_new, dtype)
            tried = True
            nsr_new = ppl_new = diff_new = 0.0
            for _ in range(num_metric_steps):
                batch = next(metric_iterator)
                tap.set_weight(w_deco)
                y_deco = module(batch)
                tap.set_weight(w0)
                y_orig = module(batch)
                l_deco, l_orig = loss_fn(batch, y_deco), loss_fn(batch, y_orig)
                nsr_s = nsr(x=y_deco, y=y_orig, non_channel_dim=(0, 1))  # dwain.py:273-275
                p_deco, p_orig = torch.exp(l_deco).mean(), torch.exp(l_orig).mean()
                diff_new += ((p_deco - p_orig) / p_orig).item()
                nsr_new += nsr_s.item()
                ppl_new += p_deco.item()
            nsr_new /= num_metric_steps
            ppl_new /= num_metric_steps
            diff_new /= num_metric_steps
            # dwain.py:460-470: three '>=' rejections in order, so a NaN metric is ACCEPTED
            accepted = not (diff_new >= thr or diff_new >= max_accepted_ppl_diff
                            or nsr_new >= nsr_final_threshold)
            if accepted:
                rank_best, nsr_best, ppl_best = rank_new, nsr_new, ppl_new
            if trace is not None:
                trace.append({"layer": name, "i": i, "rank": rank_new, "nsr": nsr_new, "ppl_deco": ppl_new,
                              "ppl_diff": diff_new, "threshold": thr, "accepted": accepted})
            i += 1

        decide = tried and is_num_params_reduced(rank_best / full, n_in, n_out)
        if tried and full != rank_best and decide:
            proportion = rank_best / full
            big_u, big_v, _ = factors(w0, u, rank_best, dtype)
            pair = build_pair(layer, big_u.T, big_v.T).to(dtype)
            drop = params_for_proportion(1.0, n_in, n_out) - params_for_proportion(proportion, n_in, n_out)
            return {"proportion": proportion, "nsr_final": nsr_best, "ppl_final": ppl_best,
                    "drop_in_params": drop, "decomposed_module": pair}
        return {"proportion": 1.0, "nsr_final": 0.0, "ppl_final": 0.0, "drop_in_params": 0,
                "decomposed_module": None}
    finally:
        tap.close()


def dwain_decompose(*, module, data_iterator, loss_fn, num_data_steps, metric_iterator, num_metric_steps,
                    nsr_final_threshold, finetune_fn=None, blacklisted_module_names=None, min_rank=32,
                    trade_off_factor=0.5, reduction_factor=0.5, max_accepted_ppl_diff=0.1,
                    decompose_in_float64=True, precomputing_covariance_num_splits=None, trace=None):
    """dwain.py:677-800 on CPU.  Returns the decompose_config dict."""
    black = blacklisted_module_names or []
    total = num_params(module)
    names = [n for n, m in module.named_modules() if is_decomposeable(m) and n not in black]
    u_dict: dict[str, torch.Tensor] = {}
    if precomputing_covariance_num_splits:
        u_dict = dwain_precompute(module, names, precomputing_covariance_num_splits, num_data_steps,
                                  data_iterator, decompose_in_float64)
    config: dict[str, Any] = {}
    done: list[str] = []
    for name in reversed(names):
        with torch.no_grad():
            res = dwain_process_layer(
                module=module, name=name, data_iterator=data_iterator, loss_fn=loss_fn,
                metric_iterator=metric_iterator, num_data_steps=num_data_steps,
                num_metric_steps=num_metric_steps, total_params=total,
                nsr_final_threshold=nsr_final_threshold, min_rank=min_rank,
                trade_off_factor=trade_off_factor, reduction_factor=reduction_factor,
                max_accepted_ppl_diff=max_accepted_ppl_diff, float64=decompose_in_float64,
                u=u_dict.pop(name) if u_dict else None, trace=trace)
        new = res["decomposed_module"]
        if new is not None:
            done.append(name)
            _set_submodule(module, name, new)
            if finetune_fn is not None:
                module = finetune_fn(module, torch.device("cpu"), done)
            cfg = module_config(new)
            cfg[META_KEY] = {k: v for k, v in res.items() if k != "decomposed_module"}
            config[name] = cfg
    return config


# --------------------------------------------------------------------------
# falor end to end
# --------------------------------------------------------------------------
def falor_process_layer(*, module, name, data_iterator, nsr_final_threshold, kl_final_threshold,
                        num_data_steps, num_metric_steps, use_float64, use_mean, use_damping, trace=None):
    """falor.py:284-399 including quirk 1: the pair is built from the LAST TRIED
    factors and nsr_final / kl_final are the last tried values."""
    layer = module.get_submodule(name)
    tap = _Tap(layer)
    try:
        w0 = tap.weight2d()
        n_out, n_in = w0.shape
        full = min(n_in, n_out)
        if full == 1:
            return {"proportion": 1.0, "nsr_final": 0.0, "kl_final": 0.0, "decomposed_module": None}
        module.eval()
        dt = torch.float64 if use_float64 else torch.float32
        ey, eyyt = torch.zeros(n_out, dtype=dt), torch.zeros((n_out, n_out), dtype=dt)
        for _ in range(num_data_steps):
            module(next(data_iterator))
            falor_accumulate(ey, eyyt, w0, tap.rows())
        _, u = falor_finalize(ey, eyyt, num_data_steps, use_mean, use_damping)

        rank_best, width = full, full // 2
        nsr_new = kl_new = 0.0
        big_u = big_v = None
        i = 1
        while width > 0:
            rank_new = rank_best - width
            big_u, big_v, w_deco = factors(w0, u, rank_new, torch.float32)
            nsr_new = kl_new = 0.0
            for _ in range(num_metric_steps):
                x = next(data_iterator)
                tap.set_weight(w_deco)
                y_deco = module(x)
                tap.set_weight(w0)
                y_orig = module(x)
                nsr_new += nsr(x=y_deco, y=y_orig, non_channel_dim=(0,)).mean().item()
                kl_new += kl_loss(y_deco, y_orig).item()
            nsr_new /= num_metric_steps
            kl_new /= num_metric_steps
            accepted = nsr_new < nsr_final_threshold and kl_new < kl_final_threshold
            if accepted:
                rank_best = rank_new
            if trace is not None:
                trace.append({"layer": name, "i": i, "width": width, "rank": rank_new, "nsr": nsr_new,
                              "kl": kl_new, "accepted": accepted})
            width //= 2
            i += 1
        tap.set_weight(w0)
        proportion = rank_best / full
        pair = None
        if full != rank_best and is_num_params_reduced(proportion, n_in, n_out):
            pair = build_pair(layer, big_u.T, big_v.T)
        return {"proportion": proportion, "nsr_final": nsr_new, "kl_final": kl_new, "decomposed_module": pair}
    finally:
        tap.close()


def falor_decompose(*, module, data_iterator, proportion_threshold, nsr_final_threshold, kl_final_threshold,
                    num_data_steps, num_metric_steps, use_float64, use_mean, use_damping,
                    blacklisted_module_names=None, trace=None):
    """falor.py:424-511 on CPU.  Returns the decompose_config dict."""
    black = blacklisted_module_names or []
    names = [n for n, m in module.named_modules() if is_decomposeable(m)]
    results = {}
    for name in names:
        if name in black:
            continue
        with torch.no_grad():
            results[name] = falor_process_layer(
                module=module, name=name, data_iterator=data_iterator,
                nsr_final_threshold=nsr_final_threshold, kl_final_threshold=kl_final_threshold,
                num_data_steps=num_data_steps, num_metric_steps=num_metric_steps,
                use_float64=use_float64, use_mean=use_mean, use_damping=use_damping, trace=trace)
    config: dict[str, Any] = {}
    for name in names:
        if name in black:
            continue
        res = results[name]
        new = res["decomposed_module"]
        if new is None or not res["proportion"] < proportion_threshold:
            continue
        _set_submodule(module, name, new)
        cfg = module_config(new)
        cfg[META_KEY] = {k: v for k, v in res.items() if k != "decomposed_module"}
        config[name] = cfg
    return config
