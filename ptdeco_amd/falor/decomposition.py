"""falor (Features Are LOw Rank) on MI355X.

Keeps the keyword API, data-consumption order, bisection decisions, quirks and return
format of ``ptdeco.falor.decompose_in_place`` (reference
src/ptdeco/falor/decomposition.py:424-511; line numbers below are in that file) while the
contractions, the eigendecomposition and the NSR / KL reductions run in HIP kernels.

Multi-GPU: pass 1 analyses every layer on the still-original model, so layers are
independent given the data stream; they are dealt round-robin to the ranks, each rank
skips the batches the other ranks' layers consume (the count is a function of the layer
shape only), and the owners broadcast the finished factors.  No matrix collective is
needed in this mode.
"""

from __future__ import annotations

import collections
import collections.abc
import logging
import time
from typing import Any, Optional

import torch

from .. import _engine as eng
from .. import utils
from ..sharding import Shard

__all__ = ["decompose_in_place", "is_decomposeable_module"]

EIGEN_DAMPEN_FACTOR = eng.EIGEN_DAMPEN_FACTOR
is_decomposeable_module = eng.is_decomposeable_module

logger = logging.getLogger(__name__)


def _bisection_widths(full_rank: int) -> list[int]:
    """:341-344, 374 -- rank_width = full // 2, halved until it reaches 0."""
    widths, w = [], full_rank // 2
    while w > 0:
        widths.append(w)
        w //= 2
    return widths


def _layer_shape(layer: torch.nn.Module) -> tuple[int, int]:
    w = layer.weight
    return w.shape[0], w.shape[1]


def _batches_consumed(layer: torch.nn.Module, num_data_steps: int, num_metric_steps: int) -> int:
    dim_out, dim_in = _layer_shape(layer)
    full = min(dim_in, dim_out)
    if full == 1:
        return 0
    return num_data_steps + num_metric_steps * len(_bisection_widths(full))


def _compute_decompositon_of_covariance_matrix(*, root_module, tap: eng.LayerTap, data_iterator, weight,
                                               num_data_steps, device, use_float64, use_mean, use_damping,
                                               top_k=None):
    """:165-208.  Quirk kept: damping is applied to Eyyt after cov was formed, so with
    use_mean=True it never reaches the matrix that is decomposed (:196-205)."""
    root_module.eval()
    with eng.phase("A_accumulate"):
        cov = eng.Covariance(weight.shape[0], device, use_float64, with_mean=True, weight=weight, top_k=top_k)
        for _ in range(num_data_steps):
            tap.calibration_forward(root_module, next(data_iterator).to(device))
            cov.add_inputs(tap.last_input_rows(), weight)
    logger.info("Using mean for covariance" if use_mean else "Not using mean for covariance")
    damp = EIGEN_DAMPEN_FACTOR if (use_damping and not use_mean) else 0.0
    with eng.phase("B_eigh"):
        return cov.eigenvectors(damp, use_mean=use_mean, top_k=top_k)


def _compute_metrics(*, x, root_module, tap: eng.LayerTap, orig_weight, candidate, key=None, pin=None) -> torch.Tensor:
    """:211-233 -- (nsr, kl) as one f64 device tensor.  `candidate` = (uk, U, W~); W~ None means
    "evaluate through the rank-r pair" (see LayerTap.use_pair)."""
    root_module.eval()
    uk, big_u, deco_weight = candidate
    if deco_weight is None:
        y_deco, y_orig = eng.forward_pair(root_module, tap, x, lambda: tap.use_pair(big_u, uk),
                                          lambda: tap.use_dense(orig_weight), key=key, pin=pin)
    else:
        y_deco, y_orig = eng.forward_pair(root_module, tap, x, lambda: tap.set_weight(deco_weight),
                                          lambda: tap.set_weight(orig_weight), key=key, pin=pin)
    nsr = utils.calc_per_channel_noise_to_signal_ratio(y=y_orig, x=y_deco, non_channel_dim=(0,))
    kl = utils.calc_kl_loss(y_deco, y_orig)
    return torch.stack([nsr, kl])


def _process_module(*, root_module, decomposed_submodule_name, data_iterator, nsr_final_threshold,
                    kl_final_threshold, num_data_steps, num_metric_steps, device, use_float64, use_mean, use_damping,
                    trace: Optional[list] = None) -> dict[str, Any]:
    """:284-399."""
    name = decomposed_submodule_name
    tap = eng.LayerTap(root_module, name)
    try:
        layer = tap.layer
        orig_weight = tap.weight_copy()
        orig_device = orig_weight.device
        dim_out, dim_in = orig_weight.shape
        full_rank = min(dim_in, dim_out)
        msg_prefix = f"Processing {name}:"
        if full_rank == 1:
            logger.info(f"{msg_prefix} Module has rank 1, not decomposing")
            return {"proportion": 1.0, "nsr_final": 0.0, "kl_final": 0.0, "decomposed_module": None}
        logger.info(f"{msg_prefix} {utils.get_type_name(layer)} weight_shape={tuple(orig_weight.shape)}")
        logger.info(f"{msg_prefix} {nsr_final_threshold=:.6f} {kl_final_threshold=:.6f}")

        # The bisection (:340-375) only ever tries ranks <= full_rank - 1, so the eigenvectors below the
        # top full_rank - 1 are never read.  For a widening layer (n_out > n_in: qkv, fc1, a classifier
        # head) that is decisive: y = x W^T has a feature covariance of rank <= n_in, i.e. an (n_out - n_in)
        # fold zero eigenvalue, and asking for its eigenvectors would push the solver to its Jacobi fallback.
        u = _compute_decompositon_of_covariance_matrix(
            root_module=root_module, tap=tap, data_iterator=data_iterator, weight=orig_weight,
            num_data_steps=num_data_steps, device=device, use_float64=use_float64, use_mean=use_mean,
            use_damping=use_damping, top_k=full_rank - 1)

        # the tapped layer runs on the HIP GEMMs while it is analysed (f32 models only: falor builds
        # its factors in float32, falor.py:346)
        fast = orig_weight.dtype == torch.float32 and tap.use_dense(orig_weight)
        tap.enable_prefix_memo(root_module)   # the two forwards of a metric step share the work ahead of this layer
        # bisection: each decision feeds the next candidate, so one host sync per candidate
        rank_best = full_rank
        nsr_best = kl_best = nsr_new = kl_new = 0.0
        uk = big_u = None
        with eng.phase("C_factors"):
            bank = eng.FactorBank(orig_weight, u, full_rank, torch.float32)  # U = W^T u once (:346-348)
        for i, rank_width in enumerate(_bisection_widths(full_rank), start=1):
            rank_new = rank_best - rank_width
            with eng.phase("C_factors"):
                candidate = bank.get(rank_new, dense=not fast)
            uk, big_u, _ = candidate
            with eng.phase("D_metrics"):
                acc = torch.zeros(2, dtype=torch.float64, device=device)
                for _ in range(num_metric_steps):
                    # (a batch that comes round again within this layer's search -- an iterator cycling over a few
                    # batches -- meets the prefix and the original output kept from its first visit: eng.PrefixMemo)
                    batch = next(data_iterator)
                    acc += _compute_metrics(x=batch.to(device), root_module=root_module, tap=tap,
                                            orig_weight=orig_weight, candidate=candidate, key=eng.batch_key(batch),
                                            pin=batch)
                nsr_new, kl_new = (acc / num_metric_steps).tolist()
                eng.warn_if_not_finite([nsr_new], decomposed_submodule_name)
            accepted = nsr_new < nsr_final_threshold and kl_new < kl_final_threshold
            if accepted:
                rank_best, nsr_best, kl_best = rank_new, nsr_new, kl_new
            logger.info(f"{msg_prefix} {i=} {rank_width=} {rank_new=} {nsr_new=:.6f} {kl_new=:.6f} "
                        f"{rank_best=} {nsr_best=:.6f} {kl_best=:.6f}")
            if trace is not None:
                trace.append({"layer": name, "i": i, "width": rank_width, "rank": rank_new, "nsr": nsr_new,
                              "kl": kl_new, "accepted": accepted})
        assert uk is not None
        tap.use_module_forward()
        tap.set_weight(orig_weight)

        proportion = rank_best / full_rank
        logger.info(f"{msg_prefix} iter=FINAL rank={rank_best} {proportion=:.4f} nsr={nsr_best:.6f} kl={kl_new:.6f}")
        new_module = None
        if full_rank != rank_best and eng.is_num_params_reduced(proportion, dim_in, dim_out):
            # quirk kept (:376-387): the pair is built from the LAST TRIED factors, and the
            # reported nsr_final / kl_final are the last tried values (:396-397)
            new_module = eng.build_pair(layer, big_u, uk, None).to(orig_device)
        else:
            logger.info(f"{msg_prefix} {proportion=:.4f} leads to num param increase, not decomposing")
        return {"proportion": proportion, "nsr_final": nsr_new, "kl_final": kl_new, "decomposed_module": new_module}
    finally:
        tap.close()


def _ship_result(result: Optional[dict[str, Any]], layer: torch.nn.Module, index: int, shard: Shard,
                 device: torch.device) -> dict[str, Any]:
    """Owner -> everyone: scalars by object broadcast, the two factor weights as tensors."""
    meta = None
    if shard.owns(index):
        pair = result["decomposed_module"]
        meta = {k: v for k, v in result.items() if k != "decomposed_module"}
        meta["_rank"] = None if pair is None else pair[0].weight.shape[0]
    meta = shard.broadcast_object(meta, index)
    r = meta.pop("_rank")
    if r is None:
        return {**meta, "decomposed_module": None}
    dim_out, dim_in = _layer_shape(layer)
    if shard.owns(index):
        pair = result["decomposed_module"]
        w1, w2 = pair[0].weight.detach(), pair[1].weight.detach()
    else:
        w1 = w2 = None
    tail = (1, 1) if isinstance(layer, torch.nn.Conv2d) else ()
    w1 = shard.broadcast_from_owner(w1, index, (r, dim_in) + tail, torch.float32, device)
    w2 = shard.broadcast_from_owner(w2, index, (dim_out, r) + tail, torch.float32, device)
    if shard.owns(index):
        return result
    flat1 = w1.reshape(r, dim_in)
    flat2 = w2.reshape(dim_out, r)
    pair = eng.build_pair(layer, flat1.T, flat2, None)
    return {**meta, "decomposed_module": pair}


def decompose_in_place(
    *,
    module: torch.nn.Module,
    device: torch.device,
    data_iterator: collections.abc.Iterator[torch.Tensor],
    blacklisted_module_names: Optional[list[str]] = None,
    proportion_threshold: float,
    nsr_final_threshold: float,
    kl_final_threshold: float,
    num_data_steps: int,
    num_metric_steps: int,
    use_float64: bool,
    use_mean: bool,
    use_damping: bool,
    process_group: Any = None,
    trace: Optional[list] = None,
) -> dict[str, Any]:
    """Same contract as ``ptdeco.falor.decompose_in_place`` (:424-511); ``process_group`` and
    ``trace`` are optional extras (see the module docstring)."""
    start_time = time.perf_counter()
    device = eng.require_device(device)
    shard = Shard.from_env(process_group)
    eng.begin_run()     # (route memory of the eigensolver: this call's own requests decide, not an earlier run's)
    blacklisted = blacklisted_module_names or []

    names = [name for name, mod in module.named_modules() if is_decomposeable_module(mod)]
    n = len(names)
    results: dict[str, dict[str, Any]] = {}
    work = [name for name in names if name not in blacklisted]
    for i, name in enumerate(names, start=1):
        if name in blacklisted:
            logger.info(f"Processing {name}: module {i} of {n}, skipped as blacklisted")
    for index, name in enumerate(work):
        layer = module.get_submodule(name)
        if shard.owns(index):
            logger.info(f"Processing {name}: module {names.index(name) + 1} of {n}")
            with torch.no_grad():
                results[name] = _process_module(
                    root_module=module, decomposed_submodule_name=name, data_iterator=data_iterator,
                    nsr_final_threshold=nsr_final_threshold, kl_final_threshold=kl_final_threshold,
                    num_data_steps=num_data_steps, num_metric_steps=num_metric_steps, device=device,
                    use_float64=use_float64, use_mean=use_mean, use_damping=use_damping, trace=trace)
        else:  # another rank's layer: keep this rank's stream position in step with the sequential order
            for _ in range(_batches_consumed(layer, num_data_steps, num_metric_steps)):
                next(data_iterator)
    if shard.active:
        for index, name in enumerate(work):
            results[name] = _ship_result(results.get(name), module.get_submodule(name), index, shard, device)

    decompose_config: dict[str, Any] = {}
    counter: collections.Counter[str] = collections.Counter()
    for name in work:
        result = results[name]
        new_module, proportion = result["decomposed_module"], result["proportion"]
        if new_module is None:
            logger.info(f"Decomposing {name}: SKIPPED {proportion=:.4f} leads to num param increase")
            continue
        if proportion < proportion_threshold:
            old_type = utils.get_type_name(module.get_submodule(name))
            utils.replace_submodule_in_place(module, name, new_module)
            module_config = utils.get_module_config(new_module)
            module_config[utils.MODCONFIG_META_KEY] = {k: v for k, v in result.items() if k != "decomposed_module"}
            decompose_config[name] = module_config
            counter[old_type] += 1
            logger.info(f"Decomposing {name}: finished {proportion=:.3f}")
        else:
            logger.info(f"Decomposing {name}: SKIPPED, {proportion=:.3f} above {proportion_threshold=:.3f}")
    for type_name, count in counter.items():
        logger.info(f"Decomposed {count} instances of {type_name}")
    logger.info(f"Total decomposable modules {n}")
    logger.info(f"Decomposition took {time.perf_counter() - start_time:.1f} seconds")
    return decompose_config
