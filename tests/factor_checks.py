"""Check of installed rank-r pairs against f64 reference arithmetic on captured calibration data -- used by the bf16
full-width test (tests/test_fullwidth_gpu.py) on every replaced layer and by the full-depth tools (tools/c4_stack.py,
tools/c4_hf_llama.py) on one replaced layer chosen at random: those runs take minutes and used to
assert nothing but "ran, N replaced".  Before the run a few layers of the FIRST precompute split are
armed -- their original weight is copied and their inputs over the calibration batches are recorded from the untouched
model (the first split's covariances are taken before anything is replaced).  After the run ONE armed layer that was
replaced, chosen at random, is checked against f64 reference arithmetic on the captured data:
  * its second factor has orthonormal columns and its first factor is (second factor)^T W  -- the pair is the projection
    of the original weight onto the span of the second factor (dwain.py:424-429);
  * that span captures as much of the layer's feature covariance C = sum_s y_s^T y_s / (T D) + damping as the r leading
    eigenvectors of C found by the library eigensolver (torch.linalg.eigh, f64) do: trace(U^T C U) against the sum of
    the r largest eigenvalues.  (An energy ratio, not a vector comparison: random-weight spectra are flat, the basis
    of a nearly degenerate invariant subspace is arbitrary and the reference's own is LAPACK's.)
Tolerances are bf16's: the factors are stored in the model dtype.  Nothing here is imported by the package."""
import random

import torch


def arm(model, names, batches, max_layers=6, seed=0):
    """Record the original weight and the inputs of up to `max_layers` of `names` over `batches` (one forward each)."""
    rng = random.Random(seed)
    picked = rng.sample(list(names), min(max_layers, len(names)))
    state = {"weights": {}, "inputs": {n: [] for n in picked}, "order": picked, "rng": rng}
    hooks = []
    for n in picked:
        mod = model.get_submodule(n)
        state["weights"][n] = mod.weight.detach().clone()
        hooks.append(mod.register_forward_pre_hook(
            lambda m, args, n=n: state["inputs"][n].append(args[0].detach().reshape(-1, args[0].shape[-1]).clone())))
    with torch.no_grad():
        for b in batches:
            model(b)
    for h in hooks:
        h.remove()
    return state


def verify(state, model, cfg, damp=0.01, name=None):
    """-> dict with the layer checked and the three figures; raises AssertionError when one is out of bounds.  `name`:
    the layer to check (default: one replaced armed layer chosen at random)."""
    replaced = [n for n in state["order"] if n in cfg]
    if not replaced:
        return {"checked": None, "note": "none of the armed layers was replaced"}
    if name is None:
        name = state["rng"].choice(replaced)
    pair = model.get_submodule(name)
    first, second = pair[0].weight.detach().double(), pair[1].weight.detach().double()   # [r, n_in], [n_out, r]
    w = state["weights"][name].double()
    r = second.shape[1]
    eye = torch.eye(r, dtype=torch.float64, device=second.device)
    orth = (second.T @ second - eye).abs().max().item()
    proj = (first - second.T @ w).norm().item() / (second.T @ w).norm().item()
    n_out = w.shape[0]
    c = torch.zeros(n_out, n_out, dtype=torch.float64, device=w.device)
    for x in state["inputs"][name]:
        y = (x @ state["weights"][name].T).double()        # the features in the model dtype, as the stand-in forms them
        c += y.T @ y / y.shape[0]
    c /= len(state["inputs"][name])
    c += torch.eye(n_out, dtype=torch.float64, device=w.device) * (damp * torch.diag(c).mean())
    lam = torch.linalg.eigvalsh(c)
    best = lam[-r:].sum().item()
    got = torch.trace(second.T @ c @ second).item()
    out = {"checked": name, "rank": r, "n_out": n_out, "orthonormality_max_dev": orth, "first_factor_rel_err": proj,
           "captured_energy_over_optimal": got / best}
    assert orth <= 2e-2, out                 # bf16 columns: 2^-9 per entry
    assert proj <= 2e-2, out
    assert got / best >= 0.99, out
    return out
