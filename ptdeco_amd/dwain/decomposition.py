"""dwain (Decomposing Weights Algorithm - an Iterative techNique) on MI355X.

Keeps the keyword API, data-consumption order, rank-search decisions and return format
of ``ptdeco.dwain.decompose_in_place`` (reference src/ptdeco/dwain/decomposition.py:677-800)
while every contraction, the eigendecomposition and the metric reductions run in the
HIP kernels of libptdeco_hip.so.  Reference line numbers below are in that file.

Multi-GPU (optional, ``torch.distributed`` initialised with the nccl = RCCL backend, one
process per GPU, identical model replica and identical data iterators on every rank):
  * calibration steps are dealt round-robin to the ranks and the partial covariance sums
    are all-reduced over xGMI -- the only collective that touches matrix data;
  * with ``precomputing_covariance_num_splits`` the eigendecompositions of a split's
    layers are owned round-robin by the ranks and broadcast when a layer's turn comes;
  * the (candidate rank, metric batch) pairs of a layer are dealt round-robin to the ranks
    (each pair reads the batch the sequential order would give it) and the three metric
    sums per candidate are all-reduced, so every rank takes the same decisions.
"""

from __future__ import annotations

import collections.abc
import logging
import os
import time
from typing import Any, Optional

import torch

from .. import _engine as eng
from .. import ops, utils
from ..sharding import Shard

__all__ = ["decompose_in_place", "is_decomposeable_module"]

EIGEN_DAMPEN_FACTOR = eng.EIGEN_DAMPEN_FACTOR
is_decomposeable_module = eng.is_decomposeable_module

logger = logging.getLogger(__name__)


def _get_params_for_proportion(proportion: float, in_features: int, out_features: int) -> int:
    """:319-330 -- parameter count of a rank (proportion * full) pair, int() truncated."""
    baseline = in_features * out_features
    proposed = (in_features + out_features) * proportion * min(in_features, out_features)
    return int(proposed) if proposed < baseline else baseline


def _candidate_ranks(full_rank: int, min_rank: int, reduction_factor: float) -> list[int]:
    """:407-408 -- geometric schedule; the last candidate may fall below min_rank."""
    ranks, r = [], full_rank
    while r > min_rank:
        r = int(r * reduction_factor)
        ranks.append(r)
    return ranks


def _get_decomposeable_submodule_names(module: torch.nn.Module, blacklisted: list[str]) -> list[str]:
    names = []
    for name, mod in module.named_modules():
        if is_decomposeable_module(mod):
            if name in blacklisted:
                logger.info(f"Skipping blacklisted module {name}")
            else:
                names.append(name)
    return names


class CovarianceComputingLinearModule(torch.nn.Module):
    """Stand-in for an nn.Linear during the all-layers precompute pass (:166-208): same
    output, and the layer's feature covariance accumulates in HBM as a side effect."""

    def __init__(self, weight: torch.nn.Parameter, bias: Optional[torch.nn.Parameter], decompose_in_float64: bool,
                 top_k: Optional[int] = None, pool: Optional[eng.SharedInputPool] = None, name: str = ""):
        super().__init__()
        if weight.dim() != 2:
            raise RuntimeError("covariance precompute supports nn.Linear only (2-D weight), like the reference "
                               "whose x @ weight.T fails for a Conv2d weight")
        self.weight = weight
        self.bias = bias
        self.name = name
        self.in_features, self.out_features = weight.shape[1], weight.shape[0]
        self.top_k = top_k  # largest rank the search can ask for; None = all eigenvectors
        # the pool decides on the first forward whether this layer keeps its own statistics or shares an
        # input moment x^T x with the other layers that read the same tensor; it sets .cov
        self.pool = pool if pool is not None else eng.SharedInputPool(0, decompose_in_float64, weight.device, "off")
        self.pool.register(self)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        rows = x.reshape(-1, self.in_features)
        y = ops.matmul(rows, self.weight.T)  # without the bias: what the covariance accumulates (:194-200)
        self.pool.observe(self, rows, y)
        out = y + self.bias if self.bias is not None else y
        return out.reshape(*x.shape[:-1], self.out_features)

    def get_eigenvectors(self) -> torch.Tensor:
        # the reference parks u on the CPU (:208); with 288 GB of HBM it stays resident
        return self.cov.eigenvectors(EIGEN_DAMPEN_FACTOR, top_k=self.top_k).to(self.weight.dtype)

    def eigen_order(self) -> tuple:
        return self.cov.eigen_order(self.top_k)

    def eigen_problem(self) -> eng.EighProblem:
        """get_eigenvectors as a problem the precompute pass can solve together with the other layers' (:162, :206-208)."""
        p = self.cov.problem(EIGEN_DAMPEN_FACTOR, top_k=self.top_k)
        inner, dtype = p.finish, self.weight.dtype
        p.finish = lambda w, v: inner(w, v).to(dtype)
        return p


def _max_candidate_rank(dim_in: int, dim_out: int, min_rank: int, reduction_factor: float) -> int:
    """Largest rank the search of :407-421 can EVALUATE (>= 1): eigenvectors below it are never read.
    Candidates that do not lower the parameter count are skipped before any factor is formed
    (:418-421) -- for a square layer that is every rank >= full / 2, so the default schedule
    (2048, 1024, ... at 4096 x 4096) needs the top 1024 eigenvectors, not 2048."""
    full = min(dim_in, dim_out)
    baseline = _get_params_for_proportion(1.0, dim_in, dim_out)
    ranks = [r for r in _candidate_ranks(full, min_rank, reduction_factor)
             if baseline - _get_params_for_proportion(r / full, dim_in, dim_out) != 0]
    return max(1, max(ranks)) if ranks else 1


def _precompute_covariance_matrix_decompositions(*, module, submodule_names, num_data_steps, data_iterator, device,
                                                 decompose_in_float64, shard: Shard, min_rank: int,
                                                 reduction_factor: float) -> dict[str, torch.Tensor]:
    """:580-633."""
    originals = {}
    pool = eng.SharedInputPool(num_data_steps, decompose_in_float64, device)
    for name in submodule_names:
        old = module.get_submodule(name)
        originals[name] = old
        logger.info(f"Replacing {name} by covariance computing wrapper")
        top_k = _max_candidate_rank(old.weight.shape[1], old.weight.shape[0], min_rank, reduction_factor)
        utils.replace_submodule_in_place(
            module, name, CovarianceComputingLinearModule(old.weight, old.bias, decompose_in_float64, top_k, pool, name))
    module.eval()
    stand_ins = [module.get_submodule(n) for n in submodule_names]
    with torch.no_grad(), eng.phase("A_accumulate"):
        for step in range(num_data_steps):
            batch = next(data_iterator)  # every rank advances the stream identically
            mine = shard.mine(step)
            if not mine and pool.discovered:
                continue
            # (the first forward also discovers which layers share an input: every rank runs it, the
            # ranks that do not own the step discard its statistics)
            pool.begin_step(accumulate=mine)
            module(utils.to_device(batch, device))
            pool.end_step()
    for group in pool.groups:
        logger.info(f"Sharing one input second moment between {group}")
    logger.info("Computing eigenvectors ...")
    u_dict: dict[str, torch.Tensor] = {}
    pending = [None] * len(stand_ins)
    if shard.active:
        with eng.phase("comm"):
            pending = pool.reduce(shard)   # per-layer sums started (same order on every rank), completed below
    pool.finalize()
    # eigendecompositions of a split are owned round-robin (non-owners receive u later); the ones
    # this rank owns are independent and run concurrently on separate streams; each first completes the
    # exchange of ITS layer's covariance sum -- the sums of the later layers travel meanwhile
    owned = [i for i in range(len(stand_ins)) if shard.owns(i)]

    def job(i):
        def run():
            if pending[i] is not None:
                pending[i]()
            return stand_ins[i].get_eigenvectors()
        return run

    if os.environ.get("PTD_EIGH_BATCHED", "1") != "0":
        # one host thread, one stream: the layers' eigenproblems grouped by order, same-order reductions batched
        def poser(i):
            def pose():
                if pending[i] is not None:
                    pending[i]()
                return stand_ins[i].eigen_problem()
            return pose

        with eng.phase("B_eigh"):
            got = eng.solve_eigenproblems(
                [poser(i) for i in owned], [stand_ins[i].eigen_order() for i in owned], device,
                costs=[eng.eigh_cost_hint(stand_ins[i].cov, stand_ins[i].out_features, stand_ins[i].top_k) for i in owned])
    else:
        # round 5's form: one host thread and one stream per chain (PTD_EIGH_BATCHED=0, kept for A/B)
        routes = [eng.eigh_route_hint(stand_ins[i].cov, stand_ins[i].out_features, stand_ins[i].top_k) for i in owned]
        costs = [eng.eigh_cost_hint(stand_ins[i].cov, stand_ins[i].out_features, stand_ins[i].top_k) for i in owned]
        with eng.phase("B_eigh"):
            got = eng.run_concurrently([job(i) for i in owned], device, routes=routes, costs=costs)
    for i, done in enumerate(pending):     # the sums this rank only contributed to: their buffers may go now
        if done is not None and not shard.owns(i):
            done()
    for name in submodule_names:
        u_dict[name] = None
    for i, u in zip(owned, got):
        u_dict[submodule_names[i]] = u
    for i, name in enumerate(submodule_names):
        if shard.active:
            with eng.phase("comm"):
                u_dict[name] = shard.broadcast_from_owner(
                    u_dict[name], i, (stand_ins[i].out_features, min(stand_ins[i].top_k, stand_ins[i].out_features)),
                    stand_ins[i].weight.dtype, device)
        logger.info(f"Replacing {name} by original linear")
        utils.replace_submodule_in_place(module, name, originals[name])
    del stand_ins
    return u_dict


def _precompute_covariance_matrix_decompositions_in_splits(*, module, modules_to_decompose, num_splits,
                                                           num_data_steps, data_iterator, device,
                                                           decompose_in_float64, shard: Shard, min_rank: int,
                                                           reduction_factor: float):
    """:636-674 -- chunks of len // num_splits layers, each chunk consumes its own data steps."""
    chunk = len(modules_to_decompose) // num_splits
    if chunk == 0:
        chunk, num_splits = 1, len(modules_to_decompose)
    parts = num_splits if len(modules_to_decompose) % num_splits == 0 else num_splits + 1
    u_dict: dict[str, torch.Tensor] = {}
    for p in range(parts):
        sub = modules_to_decompose[p * chunk:(p + 1) * chunk]
        logger.info(f"Pre computing covariance matrices for {len(sub)} modules")
        u_dict.update(_precompute_covariance_matrix_decompositions(
            module=module, submodule_names=sub, num_data_steps=num_data_steps, data_iterator=data_iterator,
            device=device, decompose_in_float64=decompose_in_float64, shard=shard, min_rank=min_rank,
            reduction_factor=reduction_factor))
    assert len(u_dict) == len(modules_to_decompose)
    return u_dict


def _compute_covariance_matrix_decomposition(*, root_module, tap: eng.LayerTap, data_iterator, weight, num_data_steps,
                                             device, decompose_in_float64, shard: Shard,
                                             top_k: Optional[int] = None, layer_index: int = 0) -> torch.Tensor:
    """:211-244 -- D model forwards, y = x W^T, Eyyt += y^T y / T, damped eigenvectors.  With several GPUs
    the partial sums are reduced to the layer's owner (rank layer_index % G), which runs the eigensolver and
    sends the top-k eigenvectors back in the weight dtype (the dtype every candidate uses them in, :424-426)."""
    root_module.eval()
    logger.info("Using float64 for decomposition" if decompose_in_float64 else "Using float32 for decomposition")
    cov = eng.Covariance(weight.shape[0], device, decompose_in_float64, weight=weight, top_k=top_k)
    tap.use_dense(weight)
    with eng.phase("A_accumulate"):
        for step in range(num_data_steps):
            batch = next(data_iterator)
            if not shard.mine(step):
                continue
            tap.calibration_forward(root_module, utils.to_device(batch, device))
            # (the layer's own forward may already have formed y = x W^T on the GPU: tap.last_features)
            cov.add_inputs(tap.last_input_rows(), weight, features=tap.last_features)
    if not shard.active:
        with eng.phase("B_eigh"):
            return cov.eigenvectors(EIGEN_DAMPEN_FACTOR, top_k=top_k)
    cov.reduce_to_owner(shard, layer_index)
    u = None
    if shard.owns(layer_index):
        with eng.phase("B_eigh"):
            u = cov.eigenvectors(EIGEN_DAMPEN_FACTOR, top_k=top_k).to(weight.dtype)
    n = weight.shape[0]
    with eng.phase("comm"):
        return shard.broadcast_from_owner(u, layer_index, (n, n if top_k is None else min(top_k, n)), weight.dtype,
                                          device)


def _compute_metrics(*, input_dict, root_module, tap: eng.LayerTap, orig_weight, candidate, loss_fn, key=None, pin=None):
    """:247-278 -- (nsr, ppl_deco, ppl_diff) as one f64 device tensor (single host sync per step).
    `candidate` = (uk, U, W~): evaluated through the rank-r pair when the layer qualifies
    (LayerTap.use_pair), else by copying W~ into the layer like the reference."""
    assert isinstance(input_dict, dict)
    root_module.eval()
    uk, big_u, deco_weight = candidate
    if deco_weight is None:
        y_deco, y_orig = eng.forward_pair(root_module, tap, input_dict, lambda: tap.use_pair(big_u, uk),
                                          lambda: tap.use_dense(orig_weight), key=key, pin=pin)
    else:
        y_deco, y_orig = eng.forward_pair(root_module, tap, input_dict, lambda: tap.set_weight(deco_weight),
                                          lambda: tap.set_weight(orig_weight), key=key, pin=pin)
    loss_deco = loss_fn(input_dict, y_deco)
    loss_orig = loss_fn(input_dict, y_orig)
    nsr = utils.calc_per_channel_noise_to_signal_ratio(y=y_orig, x=y_deco, non_channel_dim=(0, 1), mode="mean")
    ppl_deco = torch.exp(loss_deco).mean()
    ppl_orig = torch.exp(loss_orig).mean()
    ppl_diff = (ppl_deco - ppl_orig) / ppl_orig  # :445, in the loss dtype like the reference
    return torch.stack([nsr, ppl_deco.double(), ppl_diff.double()])


def _process_module(*, root_module, decomposed_submodule_name, data_iterator, loss_fn, nsr_final_threshold,
                    num_data_steps, num_metric_steps, device, metric_iterator, num_params, min_rank,
                    trade_off_factor, reduction_factor, max_accepted_ppl_diff, decompose_in_float64, u_matrix,
                    shard: Shard, trace: Optional[list] = None, layer_index: int = 0) -> dict[str, Any]:
    """:333-537."""
    indent = "    "
    name = decomposed_submodule_name
    tap = eng.LayerTap(root_module, name)
    try:
        layer = tap.layer
        orig_device, orig_dtype = layer.weight.device, layer.weight.dtype
        orig_weight = tap.weight_copy()
        dim_out, dim_in = orig_weight.shape
        full_rank = min(dim_in, dim_out)
        msg_prefix = f"Processing {name}:"
        if full_rank == 1:
            logger.info(f"{msg_prefix} Module has rank 1, not decomposing")
            return {"proportion": 1.0, "nsr_final": 0.0, "ppl_final": 0.0, "decomposed_module": None}
        logger.info(f"{msg_prefix} {utils.get_type_name(layer)} weight_shape={tuple(orig_weight.shape)} "
                    f"{orig_weight.dtype}")
        logger.info(f"{msg_prefix} {nsr_final_threshold=:.4f} {max_accepted_ppl_diff=:.4f}")

        if u_matrix is None:
            u_matrix = _compute_covariance_matrix_decomposition(
                root_module=root_module, tap=tap, data_iterator=data_iterator, weight=orig_weight,
                num_data_steps=num_data_steps, device=device, decompose_in_float64=decompose_in_float64,
                shard=shard, top_k=_max_candidate_rank(dim_in, dim_out, min_rank, reduction_factor),
                layer_index=layer_index)
            logger.info(f"Computed u_matrix, {u_matrix.dtype=}")
        else:
            logger.info(f"Using pre-computed u_matrix, {u_matrix.dtype=}")

        tap.use_module_forward()
        tap.enable_prefix_memo(root_module)   # the two forwards of a metric step share the work ahead of this layer
        # candidates that change the parameter count, in schedule order (:407-421)
        baseline_params = _get_params_for_proportion(1.0, dim_in, dim_out)
        candidates = []
        for rank_new in _candidate_ranks(full_rank, min_rank, reduction_factor):
            drop = baseline_params - _get_params_for_proportion(rank_new / full_rank, dim_in, dim_out)
            if drop == 0:
                logger.info(f"{indent}{rank_new=} does not lead to params drop, skipping")
                continue
            candidates.append((rank_new, drop))

        fast = tap.use_dense(orig_weight)  # the tapped layer runs on the HIP GEMMs while it is analysed
        # metric batches are consumed in candidate order; with several GPUs pair (c, m) is evaluated by
        # rank (c M + m) % G on exactly the batch the sequential order gives it
        sums = torch.zeros((max(len(candidates), 1), 3), dtype=torch.float64, device=device)
        # U = W^T uk once for the largest candidate; smaller ranks are column slices of it (:424-429)
        with eng.phase("C_factors"):
            bank = eng.FactorBank(orig_weight, u_matrix, candidates[0][0], orig_dtype) if candidates else None
        for c, (rank_new, _drop) in enumerate(candidates):
            batches = [next(metric_iterator) for _ in range(num_metric_steps)]
            candidate = None
            for m, batch in enumerate(batches):
                # (candidate, metric batch) pairs are dealt round-robin: finer than whole candidates, so 4 ranks
                # share 6 x 2 pairs 3/3/3/3 instead of 2/2/1/1 candidates
                if not shard.mine(c * num_metric_steps + m):
                    continue
                if candidate is None:
                    with eng.phase("C_factors"):
                        candidate = bank.get(rank_new, dense=not fast)
                with eng.phase("D_metrics"):
                    # (a batch that comes round again within this layer's search meets the prefix and the original
                    # output kept from its first visit: eng.PrefixMemo)
                    sums[c] += _compute_metrics(input_dict=utils.to_device(batch, device), root_module=root_module,
                                                tap=tap, orig_weight=orig_weight, candidate=candidate,
                                                loss_fn=loss_fn, key=eng.batch_key(batch), pin=batch)
        if shard.active:
            with eng.phase("comm"):
                shard.all_reduce_small(sums)
        with eng.phase("D_metrics"):
            table = (sums / num_metric_steps).tolist()  # the one host sync of the rank search
        eng.warn_if_not_finite([row[0] for row in table], name)

        rank_best, nsr_best, ppl_deco_best = full_rank, 0.0, 0.0
        for i, ((rank_new, drop), (nsr_new, ppl_deco_new, ppl_diff_new)) in enumerate(zip(candidates, table), 1):
            fraction_removed = drop / num_params
            ppl_diff_threshold = fraction_removed * trade_off_factor
            logger.info(f"{indent}{i=} {ppl_deco_new=:.4f} {ppl_diff_new=:.4f} {ppl_diff_threshold=:.4f} "
                        f"{fraction_removed=:.4f} {nsr_new=:.4f}")
            # :460-470 -- three '>=' tests in this order; a NaN metric passes all of them
            if ppl_diff_new >= ppl_diff_threshold:
                verdict = f"REJECTING rank {rank_new}/{full_rank} {ppl_diff_new=:.2f} >= {ppl_diff_threshold=:.2f}"
            elif ppl_diff_new >= max_accepted_ppl_diff:
                verdict = f"REJECTING rank {rank_new}/{full_rank} {ppl_diff_new=:.3f} >= {max_accepted_ppl_diff:.3f}"
            elif nsr_new >= nsr_final_threshold:
                verdict = f"REJECTING rank {rank_new}/{full_rank} {nsr_new=:.4f} >= {nsr_final_threshold=:.4f}"
            else:
                rank_best, nsr_best, ppl_deco_best = rank_new, nsr_new, ppl_deco_new
                verdict = f"ACCEPTING rank {rank_best}/{full_rank}"
            logger.info(f"{indent}{i=} {verdict}")
            if trace is not None:
                trace.append({"layer": name, "i": i, "rank": rank_new, "nsr": nsr_new, "ppl_deco": ppl_deco_new,
                              "ppl_diff": ppl_diff_new, "threshold": ppl_diff_threshold,
                              "accepted": verdict.startswith("ACCEPTING")})

        decomposition_occurred = len(candidates) > 0
        proportion = rank_best / full_rank
        decide = decomposition_occurred and eng.is_num_params_reduced(proportion, dim_in, dim_out)
        if decomposition_occurred:
            logger.info(f"{indent}i=FINAL rank={rank_best}/{full_rank} {proportion=:.4f} nsr={nsr_best:.6f} "
                        f"ppl={ppl_deco_best:.6f}")
        if decomposition_occurred and full_rank != rank_best and decide:
            with eng.phase("C_factors"):
                uk, big_u, _ = bank.get(rank_best)  # :507-511
                new_module = eng.build_pair(layer, big_u, uk, orig_dtype).to(orig_device)
            drop_in_params = baseline_params - _get_params_for_proportion(proportion, dim_in, dim_out)
            return {"proportion": proportion, "nsr_final": nsr_best, "ppl_final": ppl_deco_best,
                    "drop_in_params": drop_in_params, "decomposed_module": new_module}
        logger.info(f"{msg_prefix} Skipping module decomposition")
        return {"proportion": 1.0, "nsr_final": 0.0, "ppl_final": 0.0, "drop_in_params": 0,
                "decomposed_module": None}
    finally:
        tap.close()


def decompose_in_place(
    *,
    module: torch.nn.Module,
    device: torch.device,
    data_iterator: collections.abc.Iterator[dict[str, torch.Tensor]],
    loss_fn: collections.abc.Callable[[dict[str, torch.Tensor], torch.Tensor], torch.Tensor],
    num_data_steps: int,
    metric_iterator: collections.abc.Iterator[dict[str, torch.Tensor]],
    num_metric_steps: int,
    blacklisted_module_names: Optional[list[str]] = None,
    nsr_final_threshold: float,
    finetune_fn: collections.abc.Callable[[torch.nn.Module, torch.device, list[str]], torch.nn.Module],
    min_rank: int = 32,
    trade_off_factor: float = 0.5,
    reduction_factor: float = 0.5,
    max_accepted_ppl_diff: float = 0.1,
    decompose_in_float64: bool = True,
    precomputing_covariance_num_splits: Optional[int] = None,
    process_group: Any = None,
    trace: Optional[list] = None,
) -> dict[str, Any]:
    """Same contract as ``ptdeco.dwain.decompose_in_place`` (:677-800).

    Extra, optional: ``process_group`` (default: the world group when torch.distributed is
    initialised with more than one rank) shards the work over GPUs as described in the module
    docstring; ``trace`` receives one dict per evaluated candidate.
    """
    start_time = time.perf_counter()
    device = eng.require_device(device)
    shard = Shard.from_env(process_group)
    eng.begin_run()     # (route memory of the eigensolver: this call's own requests decide, not an earlier run's)
    num_params = utils.get_num_params(module)
    current_params = num_params
    blacklisted = blacklisted_module_names or []
    modules_to_decompose = _get_decomposeable_submodule_names(module, blacklisted)
    n = len(modules_to_decompose)
    logger.info("\n".join([f"There are {n} linear modules that can be decomposed:"]
                          + [f"  {i}. {name}" for i, name in enumerate(modules_to_decompose, 1)]))

    if precomputing_covariance_num_splits is not None and precomputing_covariance_num_splits > 0:
        u_dict = _precompute_covariance_matrix_decompositions_in_splits(
            module=module, modules_to_decompose=modules_to_decompose, num_splits=precomputing_covariance_num_splits,
            data_iterator=data_iterator, num_data_steps=num_data_steps, device=device,
            decompose_in_float64=decompose_in_float64, shard=shard, min_rank=min_rank,
            reduction_factor=reduction_factor)
    else:
        logger.info("Skipping precomputing convariance matrices")
        u_dict = {}
    # the reference calls free_gpu_reserved_memory() (gc.collect + empty_cache, ~25 ms and a
    # re-malloc of every workspace) here and after every layer (:737, 787, 795); with 288 GB of
    # HBM the caching allocator simply keeps the workspaces.

    decompose_config: dict[str, Any] = {}
    decomposed_submodules: list[str] = []
    n_decomposed = 0
    for i, name in enumerate(reversed(modules_to_decompose), start=1):
        logger.info(f"PROCESSING {name} MODULE {i} OUT OF {n}")
        with torch.no_grad():
            result = _process_module(
                root_module=module, decomposed_submodule_name=name, data_iterator=data_iterator, loss_fn=loss_fn,
                metric_iterator=metric_iterator, nsr_final_threshold=nsr_final_threshold,
                num_data_steps=num_data_steps, num_metric_steps=num_metric_steps, device=device,
                num_params=num_params, trade_off_factor=trade_off_factor, reduction_factor=reduction_factor,
                max_accepted_ppl_diff=max_accepted_ppl_diff, min_rank=min_rank,
                decompose_in_float64=decompose_in_float64,
                u_matrix=u_dict.pop(name) if len(u_dict) > 0 else None, shard=shard, trace=trace, layer_index=i)
        current_params -= result.get("drop_in_params", 0)
        logger.info(f"CURRENT PARAMS IN M: {current_params / 1e6}")
        new_module = result["decomposed_module"]
        if new_module is not None:
            decomposed_submodules.append(name)
            utils.replace_submodule_in_place(module, name, new_module)
            module = finetune_fn(module, device, decomposed_submodules)
            module_config = utils.get_module_config(new_module)
            module_config[utils.MODCONFIG_META_KEY] = {k: v for k, v in result.items() if k != "decomposed_module"}
            decompose_config[name] = module_config
            logger.info(f"{name} decomposed with rank proportion={result['proportion']:.4f}")
            n_decomposed += 1

    logger.info(f"Decomposed {n_decomposed} out of {n} modules")
    logger.info(f"Decomposition took {time.perf_counter() - start_time:.1f} seconds")
    return decompose_config
