"""Shared machinery of the dwain and falor drivers: layer taps, the covariance
accumulator that lives in HBM, factor construction and the rank-r pair builder.

Data layout in HBM (one layer at a time unless the dwain precompute pass is on):
  E     [n_out, n_out]  f64 (or f32)  lower triangle of sum_t y y^T / T, accumulated in place
  ey    [n_out]         same dtype    falor only
  C     [n_out, n_out]  f64           finalised full symmetric matrix handed to the eigensolver
  u     [n_out, n_out]  f64           eigenvectors in columns, ascending eigenvalues
  uk / U / W~           weight dtype  per candidate rank
All arithmetic goes through ptdeco_amd.ops (libptdeco_hip.so); torch is used for
allocation, views, dtype casts of slices and the user's own model forward.
"""

from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .lowrank import LowRankConv1x1, LowRankLinear, fuse_pair, warn_once

EIGEN_DAMPEN_FACTOR = 0.01  # reference dwain.py:14, falor.py:22


class PhaseTimer:
    """Device-time spans of the phases of a decomposition (A accumulate, B eigh, C factors, D metrics,
    comm), measured with HIP events on the caller's stream; bench.py installs one as ``PHASES`` for an extra,
    untimed step.  Spans include the launch gaps inside them, i.e. they add up to the step's device timeline."""

    def __init__(self):
        self.spans: list = []
        self._open: dict = {}   # name -> nesting depth: a span inside a span of the same name is not counted twice

    def span(self, name: str):
        import contextlib

        @contextlib.contextmanager
        def cm():
            depth = self._open.get(name, 0)
            self._open[name] = depth + 1
            if depth:
                try:
                    yield
                finally:
                    self._open[name] -= 1
                return
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            try:
                yield
            finally:
                e1.record()
                self._open[name] -= 1
                self.spans.append((name, e0, e1))
        return cm()

    def totals_ms(self) -> dict:
        torch.cuda.synchronize()
        out: dict = {}
        for name, e0, e1 in self.spans:
            out[name] = out.get(name, 0.0) + e0.elapsed_time(e1)
        return out


PHASES: Optional[PhaseTimer] = None


def phase(name: str):
    import contextlib

    return PHASES.span(name) if PHASES is not None else contextlib.nullcontext()


def warn_if_not_finite(nsr_values, layer_name: str) -> None:
    """ a NaN metric reads as 'accepted' in dwain's three >= tests and as 'rejected' in falor's < tests, like
    in the reference; when it is not the model's own doing (a ptd_nsr whose cached workspace was left with a raised
    ticket by an aborted launch returns NaN from then on) that is silent damage.  The cached workspaces are dropped, so
    the next call starts from an initialised one, and the event is logged once per layer."""
    import math

    if all(math.isfinite(v) for v in nsr_values):
        return
    ops.drop_cached_workspaces()
    warn_once(f"nsr-nan:{layer_name}", f"ptdeco_amd: a non-finite NSR came back while searching the rank of {layer_name}; "
              "the cached ptd_nsr workspaces were re-initialised -- if the model's own outputs are finite, rerun this layer")


def is_decomposeable_module(module: torch.nn.Module) -> bool:
    """nn.Linear, or nn.Conv2d with a 1x1 kernel and groups == 1 (dwain.py:540-546, falor.py:402-408)."""
    if isinstance(module, torch.nn.Linear):
        return True
    return (isinstance(module, torch.nn.Conv2d) and module.kernel_size[0] == 1 and module.kernel_size[1] == 1
            and module.groups == 1)


def begin_run() -> None:
    """Start of a decompose_in_place call: the eigensolver forgets the calling thread's route memory
    (ptd_eigh_forget_declines), so that two runs in one process take the same routes and give identical results."""
    from . import _hip

    if torch.cuda.is_available():
        _hip.load().ptd_eigh_forget_declines()


def require_device(device) -> torch.device:
    device = torch.device(device)
    if device.type != "cuda":
        raise ValueError(
            f"ptdeco_amd runs the decomposition path on an MI355X; got device={device}. There is no CPU path.")
    return device


def _flat_tensors(obj, out: list) -> bool:
    """Collects the tensors of a nest of tuples / lists / dicts; False when anything else than tensors, None and
    plain scalars sits in it (a cache object, a module, ...)."""
    if isinstance(obj, torch.Tensor):
        out.append(obj)
        return True
    if obj is None or isinstance(obj, (bool, int, float, str, torch.dtype, torch.device, torch.Size)):
        return True
    if isinstance(obj, (tuple, list)):
        return all(_flat_tensors(o, out) for o in obj)
    if isinstance(obj, dict):
        return all(isinstance(k, str) and _flat_tensors(v, out) for k, v in obj.items())
    return False


def _container_print(v):
    """A cheap fingerprint of a mutable container bound to a module attribute: its length and the identities of (up to
    64 of) its elements -- an in-place `append`, `pop`, `update` or `clear` keeps the container's own id."""
    import collections

    if isinstance(v, (list, collections.deque)):
        return (len(v), tuple(id(o) for o in list(v)[:64]))
    if isinstance(v, dict):
        return (len(v), tuple((id(k), id(o)) for k, o in list(v.items())[:64]))
    if isinstance(v, (set, bytearray)):
        return (len(v),)
    return None


def _rng_state() -> tuple:
    """The generators a forward could draw from without it showing in its arguments: Python's and numpy's global ones
    and torch's default CPU / current-device generators (hashes of the states: compared, never restored)."""
    import random

    st = [hash(random.getstate())]
    try:
        import sys

        np = sys.modules.get("numpy")
        if np is not None:
            st.append(hash(np.random.get_state()[1].tobytes()))
    except Exception:
        pass
    st.append(hash(torch.get_rng_state().numpy().tobytes()))
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        gen = torch.cuda.default_generators[torch.cuda.current_device()]
        st.append((gen.initial_seed(), int(gen.get_offset()) if hasattr(gen, "get_offset") else 0))
    return tuple(st)


def _subtree_state(mod: torch.nn.Module) -> list:
    """What a forward that is a function of its arguments leaves alone: training flags, which objects the modules'
    attributes are bound to, the contents of list / dict / set attributes (fingerprints: an in-place append keeps the
    binding), the version counters of parameters and buffers, and the random generators of the process."""
    st: list = [_rng_state()]
    for m in mod.modules():
        st.append(m.training)
        for k, v in m.__dict__.items():
            if k not in ("_parameters", "_buffers", "_modules"):
                st.append((k, id(v), _container_print(v)))
        for d in (m._parameters, m._buffers):
            for k, v in d.items():
                st.append((k, id(v), -1 if v is None or v.is_inference() else v._version))
    return st


class PrefixMemo:
    """One metric step runs the model twice on the same input -- the candidate, then the original (dwain.py:263-267,
    falor.py:223-227) -- and the two forwards differ in the tapped layer alone: everything the model computes BEFORE
    that layer's first call is computed twice with the same operands.  The memo keeps, during the first forward, what
    ran before the tapped layer and hands it back in the second.  Nothing outlives the step (no reuse across batches,
    candidates or calls) and the numbers are the ones a second run on the same operands gives.

    Units.  (1) Matrix-product modules (Linear / conv / fused rank-r pairs): always.  (2) Whole SUBTREES of the module
    tree that do not contain the tapped layer -- the siblings along the path from the root to it: the decoder blocks
    in front of the block under analysis, the attention block in front of an MLP layer -- so that their norms,
    activations and attention are skipped as well and one output per block is kept instead of one per product.  A
    subtree qualifies only if it is seen to be a function of its arguments: no forward hooks inside it (they would not
    fire), eval mode, arguments and results that are nests of tensors / None / scalars (a cache object passed in
    disqualifies it), and a first recorded call that leaves every attribute binding, parameter and buffer version of
    the subtree as it found them; otherwise its products are kept one by one as in (1).
    ``PTD_PREFIX_MEMO_UNITS=products`` keeps to (1).

    Across candidates (round 6; SURVEY 8f-2 "share the y_orig forward across candidates").  The rank search of one layer
    evaluates candidate after candidate on metric batches that come round again (an iterator cycling over a few
    batches), and between two candidates of a layer nothing of the model changes but the tapped layer: what ran ahead of
    it for batch b is the same for every candidate, and so is the ORIGINAL model's output on b.  With a batch key
    (`batch_key`: the identities, version counters and addresses of the batch's tensors, a sampled fingerprint of host
    tensors; the batch object is pinned while its entries live) the prefix outputs of a batch are kept for the layer's
    whole search (`first(key)` replays them for the next candidate) and the original output is kept beside them
    (`orig_get` / `orig_put`: the second forward of a pair is not run at all).  The first replay of either kind in a
    memo's life is recomputed and compared like the in-pair self-check.  At most half the byte budget holds such
    persistent entries (PTD_MEMO_KEYS batches, default 64); a batch that does not fit is handled as before (kept for
    its own pair only).  Everything is dropped when the layer's search ends (`close`).  PTD_MEMO_ACROSS_CANDIDATES=0:
    nothing outlives a pair.

    Guards: recording stops at the tapped layer's first call and when the byte budget is spent; entries are keyed by
    the call index of their module within the forward (weight sharing, a module called again behind the tapped layer);
    a kept tensor that was modified in place since (``relu_`` on a conv output: the version counter), an inference
    tensor (no counter) or a call that sees another input shape is recomputed; modules that already carry an
    instance-level ``forward`` are left alone.  ``PTD_PREFIX_MEMO_MB`` = byte budget in MiB (default 8192, 0 = off),
    ``PTD_PREFIX_MEMO_CHECK=1`` recomputes every kept output and raises on a difference beyond the
    rounding of its dtype (tests)."""

    IDLE, RECORD, REPLAY = 0, 1, 2
    UNKNOWN, PURE, IMPURE, RANDOM = 0, 1, 2, 3     # RANDOM: the subtree drew from a global random generator -- what runs
                                                   # behind it sees other values in the second forward: recording stops there
    total_hits = 0      # outputs handed back since the process started (tests, tools)
    total_orig_hits = 0   # second forwards not run: the original output of the batch was kept from an earlier candidate
    # Self-check (round 5): the FIRST metric step of every layer recomputes each output it is about to hand back and
    # compares (one extra partial forward per layer).  A mismatch -- a model that is not a function of its input ahead
    # of the layer in a way the purity test did not see -- costs nothing but speed: the recomputed value is used, one
    # WARNING is logged, and the memo is off for this model from then on (level[id(root)] = 2; 1 = products only is
    # reserved for PTD_PREFIX_MEMO_UNITS=products).
    level: dict = {}

    def __init__(self, root: torch.nn.Module, layer: torch.nn.Module, budget_bytes: int, check: bool = False,
                 subtrees: bool = True, self_check: bool = False):
        self.mode = self.IDLE
        self.reached = False
        self.full = False
        self.bytes = 0
        self.budget = budget_bytes
        self.check = check
        import weakref

        self.root_id = id(root)
        self._root_ref = weakref.ref(root)
        self.self_check = self_check   # the first `second()` of this memo's life compares; cleared after it
        self.disabled = PrefixMemo.level.get(self.root_id, 0) >= 2
        if PrefixMemo.level.get(self.root_id, 0) >= 1:
            subtrees = False
        self.hits = 0
        self.unit_hits = 0
        self._gen = 0            # forward counter
        import collections
        import os

        self._key = None         # store of the running forward: a batch key, or ("pair", n) for a pair of its own
        self._stores: "collections.OrderedDict" = collections.OrderedDict()   # key -> {as_unit, bytes, persistent, done, pin, slots}
        self._pending = None     # a non-persistent store whose pair has not released it yet
        self.persist_bytes = 0
        self.across = os.environ.get("PTD_MEMO_ACROSS_CANDIDATES", "1") != "0"
        self.max_keys = max(0, int(os.environ.get("PTD_MEMO_KEYS", "64")))
        self._verify = False     # this forward recomputes what it is about to hand back
        self.cross_check = self_check    # the first replay in a FIRST forward (another candidate's prefix) compares
        self.orig_check = self_check     # the first original output handed back compares
        self._orig: dict = {}    # batch key -> (output nest, its tensors, their versions, pinned batch)
        self.orig_hits = 0
        self.prefix_replays = 0  # first forwards that ran on another candidate's prefix
        self._seq = 0            # calls that reached a patched forward at depth 0 in this forward: the entries' keys
        self._as_unit: set = set()   # keys of the calls that ran a subtree as ONE unit while recording
        self._depth = 0          # > 0 while a subtree runs as a unit: the patched modules inside it are transparent
        self._patched: list = []   # (module, kept dict)
        self._units: list = []     # (module, kept dict, is a subtree unit, [purity])
        import torch.nn.modules.module as _mm

        if getattr(_mm, "_global_forward_hooks", None) or getattr(_mm, "_global_forward_pre_hooks", None):
            subtrees = False
        seen: set = set()

        def products(mod):
            fused_below: list = []
            for name, m in mod.named_modules():
                if m is layer or id(m) in seen or any(name.startswith(p) for p in fused_below):
                    continue
                if isinstance(m, (LowRankLinear, LowRankConv1x1)):
                    fused_below.append(name + "." if name else "")
                elif not isinstance(m, (torch.nn.Linear, torch.nn.modules.conv._ConvNd)):
                    continue
                if "forward" in m.__dict__:
                    continue
                seen.add(id(m))
                self._patch(m, unit=False)

        def is_product(m):
            return isinstance(m, (LowRankLinear, LowRankConv1x1, torch.nn.Linear, torch.nn.modules.conv._ConvNd))

        def hooks_inside(mod):
            return any(c._forward_hooks or c._forward_pre_hooks or c._backward_hooks or "forward" in c.__dict__
                       for c in mod.modules() if c is not mod)

        def off_path(mod):
            # a module beside the path to the tapped layer: its products, and the module as a whole if it qualifies
            if isinstance(mod, (torch.nn.ModuleList, torch.nn.ModuleDict)):   # (no forward of their own)
                for c in mod.children():
                    off_path(c)
                return
            unit_ok = (subtrees and not is_product(mod) and id(mod) not in seen and "forward" not in mod.__dict__
                       and any(True for _ in mod.children()) and not hooks_inside(mod))
            products(mod)
            if unit_ok:
                seen.add(id(mod))
                self._patch(mod, unit=True)

        def on_path(mod):
            for c in mod.children():
                if c is layer:
                    continue
                if any(m is layer for m in c.modules()):
                    on_path(c)
                else:
                    off_path(c)

        if any(m is layer for m in root.modules()):
            on_path(root)
        else:
            products(root)

    @staticmethod
    def from_env(root: torch.nn.Module, layer: torch.nn.Module) -> Optional["PrefixMemo"]:
        import os

        mb = float(os.environ.get("PTD_PREFIX_MEMO_MB", "8192"))
        if mb <= 0:
            return None
        return PrefixMemo(root, layer, int(mb * 2**20), check=os.environ.get("PTD_PREFIX_MEMO_CHECK", "0") == "1",
                          subtrees=os.environ.get("PTD_PREFIX_MEMO_UNITS", "subtrees") != "products",
                          self_check=os.environ.get("PTD_PREFIX_MEMO_SELF_CHECK", "1") != "0")

    @staticmethod
    def _first_shape(args, kwargs):
        flat: list = []
        _flat_tensors(args, flat)
        _flat_tensors(kwargs, flat)
        return flat[0].shape if flat else None

    def _step_down(self, m, unit: bool) -> None:
        """A kept output differed from its recomputation: what runs behind this module sees other values in the second
        forward than in the first, so nothing may be handed back any more -- the memo is off for this model, now and
        for the memos of its later layers (level 2; only a mismatch can tell a forward whose VALUES change from one
        that merely has side effects, and nothing short of off is safe for the former)."""
        import logging

        PrefixMemo.level[self.root_id] = 2
        root = self._root_ref()
        if root is not None:
            # (keyed by id: the entry goes with the model, so that a later model at the same address does not inherit it)
            import weakref

            weakref.finalize(root, PrefixMemo.level.pop, self.root_id, None)
        logging.getLogger(__name__).warning(
            "ptdeco_amd: the model does not compute the same values ahead of the analysed layer in both forwards of a "
            "metric step (%s %s): the prefix memo is off for this model -- both forwards of every metric step run whole, "
            "as in the reference", "subtree" if unit else "module", type(m).__name__)
        for mod, kept, is_unit, purity in self._units:
            purity[0] = self.IMPURE
            kept.clear()
        self.disabled = True

    def _same(self, again, out) -> bool:
        fa: list = []
        fb: list = []
        same = _flat_tensors(again, fa) and _flat_tensors(out, fb) and len(fa) == len(fb)
        for x, y in zip(fa, fb):
            if not same:
                break
            if x.shape != y.shape or x.dtype != y.dtype:
                same = False
            elif not torch.equal(x, y):
                tol = {torch.float64: 1e-10, torch.float32: 1e-4}.get(y.dtype, 2e-2)
                same = bool((x.double() - y.double()).abs().max() <= tol * y.double().abs().max())
        return same

    def _compare(self, m, again, out) -> None:
        # (bit-identical for this package's kernels; a library kernel may vary from call to call on identical input --
        # one MIOpen 3x3 stride-2 convolution of the ResNet-18 test does, by an ulp -- so the bound is the rounding of
        # the dtype, far below what a forward that is not a function of its input -- dropout, running statistics --
        # would show)
        fa: list = []
        fb: list = []
        same = _flat_tensors(again, fa) and _flat_tensors(out, fb) and len(fa) == len(fb)
        for x, y in zip(fa, fb):
            if not same:
                break
            if x.shape != y.shape or x.dtype != y.dtype:
                same = False
            elif not torch.equal(x, y):
                tol = {torch.float64: 1e-10, torch.float32: 1e-4}.get(y.dtype, 2e-2)
                same = bool((x.double() - y.double()).abs().max() <= tol * y.double().abs().max())
        if not same:
            raise RuntimeError(f"ptdeco_amd: prefix memo mismatch in {type(m).__name__}: the model does not compute "
                               "the same values before the analysed layer in both forwards of a metric step")

    def _patch(self, m: torch.nn.Module, unit: bool) -> None:
        inner = m.forward
        kept: dict = {}
        purity = [self.UNKNOWN if unit else self.PURE]

        def whole(args, kwargs):
            # the module run as ONE unit: the patched modules inside it neither keep, replay nor count calls
            self._depth += 1
            try:
                return inner(*args, **kwargs)
            finally:
                self._depth -= 1

        def forward(*args, **kwargs):
            if self.mode == self.IDLE or self._depth > 0:
                return inner(*args, **kwargs)
            # Entries are keyed by the position of the call among the calls that reach this point in the forward (ADVICE
            # r4: per-module call indices shifted when a replayed subtree skipped the calls inside it -- a weight-tied
            # Linear called once inside a subtree and again behind it could be handed the wrong output).  Both forwards
            # count the same calls: what ran as one unit while recording runs as one unit in the replay too.
            idx = self._seq
            self._seq += 1
            if self.mode == self.RECORD:
                if purity[0] == self.RANDOM:
                    self.full = True          # nothing from here on is kept in this forward
                keep = not (self.reached or self.full) and purity[0] != self.IMPURE
                if not keep:
                    return inner(*args, **kwargs)
                if unit:
                    flat_in: list = []
                    if not (_flat_tensors(args, flat_in) and _flat_tensors(kwargs, flat_in)):
                        purity[0] = self.IMPURE
                        return inner(*args, **kwargs)
                    if purity[0] == self.UNKNOWN and any(c.training for c in m.modules()):
                        purity[0] = self.IMPURE
                        return inner(*args, **kwargs)
                    before = _subtree_state(m) if purity[0] == self.UNKNOWN else None
                    self._as_unit.add(idx)
                    out = whole(args, kwargs)
                    if before is not None:
                        after = _subtree_state(m)
                        purity[0] = self.PURE if before == after else (self.RANDOM if before[0] != after[0] else self.IMPURE)
                        if purity[0] == self.RANDOM:
                            self.full = True
                    if purity[0] != self.PURE or self.reached:    # (the tapped layer ran inside: not a prefix unit)
                        return out
                else:
                    out = inner(*args, **kwargs)
                    if self.reached:
                        return out
                flat: list = []
                if not _flat_tensors(out, flat) or not flat or any(t.is_inference() for t in flat):
                    if unit:
                        purity[0] = self.IMPURE
                    return out
                nbytes = sum(t.numel() * t.element_size() for t in {id(t): t for t in flat}.values())
                if self.bytes + nbytes <= self.budget:
                    kept[(self._key, idx)] = (out, flat, [t._version for t in flat], self._first_shape(args, kwargs))
                    self.bytes += nbytes
                    store = self._stores[self._key]
                    store["bytes"] += nbytes
                    store["slots"].append((kept, (self._key, idx)))
                    if store["persistent"] and self.persist_bytes + store["bytes"] > self.budget // 2:
                        store["persistent"] = False      # (does not fit the persistent half: this pair only)
                else:
                    self.full = True
                return out
            # REPLAY
            as_unit = unit and idx in self._as_unit
            entry = kept.get((self._key, idx))
            if entry is not None:
                out, flat, versions, shape = entry
                if all(t._version == v for t, v in zip(flat, versions)) and self._first_shape(args, kwargs) == shape:
                    if self.check:
                        self._compare(m, whole(args, kwargs) if as_unit else inner(*args, **kwargs), out)
                    elif self._verify:
                        again = whole(args, kwargs) if as_unit else inner(*args, **kwargs)
                        if not self._same(again, out):
                            self._step_down(m, unit)
                            return again
                    self.hits += 1
                    self.unit_hits += 1 if unit else 0
                    PrefixMemo.total_hits += 1
                    return out
            return whole(args, kwargs) if as_unit else inner(*args, **kwargs)

        m.forward = forward
        self._patched.append((m, kept))
        self._units.append((m, kept, unit, purity))

    def _drop(self) -> None:
        for _, kept in self._patched:
            kept.clear()
        self._stores.clear()
        self._orig.clear()
        self._pending = None
        self.bytes = 0
        self.persist_bytes = 0

    def _release(self, key) -> None:
        store = self._stores.pop(key, None)
        if store is None:
            return
        for kept, slot in store["slots"]:
            kept.pop(slot, None)
        self.bytes -= store["bytes"]
        if store["done"] and store["persistent"]:
            self.persist_bytes -= store["bytes"]

    def first(self, key=None, pin=None):
        """Context of the first forward of a metric step: outputs are kept -- or, with a batch `key` whose prefix an
        earlier candidate recorded, handed back."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            if self._pending is not None:       # (a pair that ended without its second forward)
                self._release(self._pending)
                self._pending = None
            self.reached = self.full = False
            self._gen += 1
            self._seq = 0
            self._depth = 0
            k = key if (self.across and key is not None) else None
            store = self._stores.get(k) if k is not None else None
            if self.disabled:
                self.mode = self.IDLE
            elif store is not None and store["done"] and store["persistent"]:
                self._key, self._as_unit = k, store["as_unit"]
                self.mode = self.REPLAY
                self._verify = self.cross_check
                self.prefix_replays += 1
            else:
                if store is not None:
                    self._release(k)
                self._key = k if k is not None else ("pair", self._gen)
                self._as_unit = set()
                self._stores[self._key] = {"as_unit": self._as_unit, "bytes": 0, "done": False, "pin": pin, "slots": [],
                                           "persistent": k is not None and len(self._stores) < self.max_keys}
                self.mode = self.RECORD
            try:
                yield
            finally:
                if self.mode == self.REPLAY:
                    self.cross_check = False
                elif self.mode == self.RECORD:
                    store = self._stores.get(self._key)
                    if store is not None and not store["done"]:
                        store["done"] = True
                        if store["persistent"]:
                            self.persist_bytes += store["bytes"]
                        else:
                            self._pending = self._key
                self._verify = False
                self.mode = self.IDLE
        return cm()

    def second(self, key=None):
        """Context of the second forward: kept outputs are handed back; what was kept for this pair alone is released."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            self._gen += 1
            self._seq = 0
            self._depth = 0
            store = self._stores.get(self._key)
            if self.disabled or store is None:
                self.mode = self.IDLE
            else:
                self._as_unit = store["as_unit"]
                self.mode = self.REPLAY
                self._verify = self.self_check
            try:
                yield
            finally:
                self.mode = self.IDLE
                self.self_check = False
                self._verify = False
                if self._pending is not None:
                    self._release(self._pending)
                    self._pending = None
        return cm()

    # the ORIGINAL model's output on a batch, kept for the other candidates of the layer (see the class comment)
    def orig_get(self, key):
        if not self.across or self.disabled or key is None:
            return None
        entry = self._orig.get(key)
        if entry is None:
            return None
        out, flat, versions, _pin = entry
        if not all(t._version == v for t, v in zip(flat, versions)):
            self._orig.pop(key, None)
            return None
        return out

    def orig_put(self, key, out, pin=None) -> None:
        if not self.across or self.disabled or key is None or key in self._orig or len(self._orig) >= self.max_keys:
            return
        flat: list = []
        if not _flat_tensors(out, flat) or not flat or any(t.is_inference() for t in flat):
            return
        nbytes = sum(t.numel() * t.element_size() for t in {id(t): t for t in flat}.values())
        if self.bytes + nbytes > self.budget:
            return
        self.bytes += nbytes
        self._orig[key] = (out, flat, [t._version for t in flat], pin)

    def close(self) -> None:
        self._drop()
        for m, _ in self._patched:
            m.__dict__.pop("forward", None)
        self._patched = []


def batch_key(batch):
    """Identity of a batch as the data iterator handed it over: for every tensor in it (id, version counter, address, shape,
    dtype) and, for host tensors, 64 sampled values (a buffer refilled from outside torch -- numpy writing into
    `torch.from_numpy` memory -- does not bump the counter).  None when the batch holds anything but tensors and plain
    values.  The caller pins the batch object while it uses the key, so an id cannot come round on another tensor."""
    parts: list = []
    seen_tensor = [False]

    def walk(obj) -> bool:
        if isinstance(obj, torch.Tensor):
            t = obj
            if t.is_inference():
                return False
            seen_tensor[0] = True
            part = (id(t), t._version, t.data_ptr(), tuple(t.shape), t.dtype, t.device.type)
            if t.device.type == "cpu" and t.numel() > 0 and t.layout == torch.strided and not t.is_complex():
                v = t.detach().reshape(-1) if t.is_contiguous() else t.detach().flatten()
                sample = v[:: max(1, v.numel() // 64)][:64]
                part += (tuple(sample.to(torch.float64).tolist()),)
            parts.append(part)
            return True
        if obj is None or isinstance(obj, (bool, int, float, str, torch.dtype, torch.device, torch.Size)):
            parts.append(("value", obj))      # (a plain value the model may read: part of the batch's identity)
            return True
        if isinstance(obj, (tuple, list)):
            parts.append(("seq", len(obj)))
            return all(walk(o) for o in obj)
        if isinstance(obj, dict):
            parts.append(("dict", tuple(obj.keys()) if all(isinstance(k, str) for k in obj) else None))
            return all(isinstance(k, str) and walk(v) for k, v in obj.items())
        return False

    if not walk(batch) or not seen_tensor[0]:
        return None
    return tuple(parts)


def forward_pair(root: torch.nn.Module, tap: "LayerTap", x, first_setup, second_setup, key=None, pin=None):
    """The two forwards of one metric step: ``first_setup()``, model, ``second_setup()``, model -- the second reusing
    what the first computed ahead of the tapped layer (PrefixMemo) when the tap carries a memo.  With a batch `key`
    (batch_key of the batch `pin` as the iterator yielded it) the first forward runs on the prefix an earlier candidate
    of this layer recorded for the same batch, and the second -- the original model on that batch, the same for every
    candidate -- is run once per batch and layer."""
    import contextlib

    memo = tap.memo
    first_setup()
    with memo.first(key, pin) if memo is not None else contextlib.nullcontext():
        y_first = root(x)
    cached = memo.orig_get(key) if memo is not None else None
    if cached is not None and not memo.orig_check:
        memo.orig_hits += 1
        PrefixMemo.total_orig_hits += 1
        return y_first, cached
    second_setup()
    with memo.second(key) if memo is not None else contextlib.nullcontext():
        y_second = root(x)
    if memo is not None and key is not None:
        if cached is not None:      # the first original output about to be handed back in this memo's life: compared
            memo.orig_check = False
            if not memo._same(y_second, cached):
                memo._step_down(root, True)
        else:
            memo.orig_put(key, y_second, pin)
    return y_first, y_second


class _StopForward(BaseException):
    """Raised by a LayerTap when the calibration forward has delivered the analysed layer's input (a BaseException:
    `except Exception` in the caller's model does not swallow it)."""


class LayerTap:
    """Records the last input of one decomposable layer and gives a 2-D view of its weight.

    Stands in for the reference's wrapper modules (dwain.py:41-144, falor.py:51-153):
    a forward pre-hook instead of swapping the module, so the model's module tree and
    parameter names are untouched while a layer is being analysed.
    """

    def __init__(self, root: torch.nn.Module, name: str):
        layer = root.get_submodule(name)
        if not is_decomposeable_module(layer):
            raise ValueError(f"Cannot decompose {name}={layer}")
        self.name = name
        self.layer = layer
        self.is_conv = isinstance(layer, torch.nn.Conv2d)
        self._last: Optional[torch.Tensor] = None
        self.last_features: Optional[torch.Tensor] = None  # [T, n_out] of the last use_dense forward
        self.memo: Optional[PrefixMemo] = None
        self.calls = 0                 # calls of the layer since calibration_forward last reset it
        self._stop_at_input = False
        self._single_call_seen = False
        self._cal_forwards = 0
        self._plain_ancestors: Optional[bool] = None
        self._handle = layer.register_forward_pre_hook(self._record)

    def _record(self, _module, args) -> None:
        self._last = args[0]
        self.calls += 1
        if self.memo is not None:
            self.memo.reached = True
        if self._stop_at_input:
            self.last_features = None
            raise _StopForward()

    def calibration_forward(self, root: torch.nn.Module, batch) -> None:
        """One forward of the model whose only purpose is this layer's input (dwain.py:236-239, falor.py:189-193: the
        model's output is discarded).  The first such forward runs to the end and counts the layer's calls; if it was
        called exactly once, the later ones stop at the layer -- everything behind it would be computed for nothing
        (at depth f of a stack 1 - f of the forward).  A layer that is called several times per forward keeps the
        full forwards: the reference uses the LAST call's input (get_last_input).  PTD_CALIBRATION_EARLY_STOP=0
        keeps every forward whole."""
        import os

        self.calls = 0
        self._cal_forwards += 1
        #  the call count of ONE batch is not a fact about every batch (data-dependent control flow), so every
        # 8th forward runs whole again and re-establishes it; and the unwinding exception skips whatever the modules
        # around the layer do behind their inner forward: with a hook or an instance-level forward on any ancestor
        # (accelerate's offload hooks are such wrappers) the forwards stay whole
        revalidate = self._cal_forwards % 8 == 0
        if revalidate:
            self._plain_ancestors = None      # (a hook registered on an ancestor since the last look is seen)
        if (self._single_call_seen and not revalidate and self._ancestors_plain(root)
                and os.environ.get("PTD_CALIBRATION_EARLY_STOP", "1") != "0"):
            self._stop_at_input = True
            try:
                root(batch)
            except _StopForward:
                pass
            finally:
                self._stop_at_input = False
            return
        root(batch)
        self._single_call_seen = self.calls == 1

    def _ancestors_plain(self, root: torch.nn.Module) -> bool:
        """No module on the path from the root to the layer carries hooks or an instance-level forward."""
        if self._plain_ancestors is None:
            ok = True
            parts = self.name.split(".")
            mod = root
            for part in [None] + parts[:-1]:
                if part is not None:
                    mod = getattr(mod, part)
                if (mod._forward_hooks or mod._forward_pre_hooks or mod._backward_hooks or "forward" in mod.__dict__
                        or getattr(mod, "_hf_hook", None) is not None):
                    ok = False
                    break
            self._plain_ancestors = ok
        return self._plain_ancestors

    def enable_prefix_memo(self, root: torch.nn.Module) -> None:
        """Metric steps of this layer's rank search share the model's work ahead of the layer (PrefixMemo)."""
        if self.memo is None:
            self.memo = PrefixMemo.from_env(root, self.layer)

    def close(self) -> None:
        self.use_module_forward()
        if self.memo is not None:
            self.memo.close()
            self.memo = None
        self._handle.remove()
        self._last = None

    @property
    def n_in(self) -> int:
        return self.layer.in_channels if self.is_conv else self.layer.in_features

    def last_input_rows(self) -> torch.Tensor:
        """[T, n_in] rows of the last input (NCHW -> NHWC rows for a conv: dwain.py:64, 116)."""
        x = self._last
        if x is None:
            raise RuntimeError(f"layer {self.name} was not reached by the model's forward")
        if self.is_conv:
            return x.permute(0, 2, 3, 1).reshape(-1, self.n_in)
        return x.reshape(-1, self.n_in)

    def weight_copy(self) -> torch.Tensor:
        w = self.layer.weight.detach()
        return (w[..., 0, 0] if self.is_conv else w).clone()

    def set_weight(self, w2d: torch.Tensor) -> None:
        if self.is_conv:
            self.layer.weight.copy_(w2d[:, :, None, None])
        else:
            self.layer.weight.copy_(w2d)

    # -- the tapped layer's own forward on the HIP kernels ---------------------------------
    # The reference evaluates a candidate by copying W~ = (U V)^T into the live layer and
    # running the model, then copying W back and running it again (dwain.py:263-267,
    # falor.py:223-227).  The same two functions are evaluated here without touching the
    # weight: the candidate through its rank-r pair (x U) uk^T + b -- exactly what will be
    # deployed, 2 T r (n_in + n_out) instead of 2 T n_in n_out flops and no W~ product -- and
    # the original through the f32/bf16 MFMA GEMM.  Only plain Linear / stride-1 1x1 conv
    # layers in f32 or bf16 qualify; anything else keeps the module's own forward.
    def _plain(self) -> bool:
        l = self.layer
        if l.weight.dtype not in (torch.float32, torch.bfloat16) or not l.weight.is_cuda:
            return False
        if self.is_conv:
            return (tuple(l.stride) == (1, 1) and tuple(l.dilation) == (1, 1) and l.padding_mode == "zeros"
                    and l.padding in ((0, 0), 0, "valid"))
        return True

    def _rows_in(self, x: torch.Tensor) -> torch.Tensor:
        return x.permute(0, 2, 3, 1).reshape(-1, self.n_in) if self.is_conv else x.reshape(-1, self.n_in)

    def _rows_out(self, y: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        if self.is_conv:
            b, _, h, w = x.shape
            return y.reshape(b, h, w, y.shape[-1]).permute(0, 3, 1, 2)
        return y.reshape(*x.shape[:-1], y.shape[-1])

    def use_dense(self, weight2d: torch.Tensor) -> bool:
        """Layer forward = x weight2d^T + bias on the matrix cores.  Returns False if unsupported."""
        if not self._plain():
            return False
        bias = self.layer.bias

        def fwd(x):
            if x.dtype != weight2d.dtype:  # e.g. autocast: not a HIP GEMM input, say so and let torch run the layer
                warn_once(f"tap-dense:{x.dtype}", f"ptdeco_amd: layer {self.name} received {x.dtype} inputs for "
                          f"{weight2d.dtype} weights; its forward stays in torch for this run")
                self.last_features = None
                return type(self.layer).forward(self.layer, x)
            y = ops.matmul(self._rows_in(x), weight2d.T)
            self.last_features = y  # x W^T without bias: what the covariance accumulates
            return self._rows_out(y + bias if bias is not None else y, x)

        self.layer.forward = fwd
        return True

    def use_pair(self, big_u: torch.Tensor, uk: torch.Tensor) -> bool:
        """Layer forward = (x U) uk^T + bias (the rank-r candidate).  Returns False if unsupported."""
        if not self._plain():
            return False
        first = big_u.T.contiguous()  # [r, n_in]
        bias = self.layer.bias

        def fwd(x):
            if x.dtype != first.dtype:
                raise TypeError(f"ptdeco_amd: layer {self.name} received {x.dtype} inputs but its candidate factors are "
                                f"{first.dtype}; the rank-r candidate cannot be evaluated through the pair")
            return self._rows_out(ops.lowrank_forward(self._rows_in(x), first, uk, bias), x)

        self.layer.forward = fwd
        return True

    def use_module_forward(self) -> None:
        self.layer.__dict__.pop("forward", None)
        self.last_features = None


def _input_route_wanted(n_out: int, n_in: int, top_k: Optional[int]) -> bool:
    """Layers that widen their input (n_out > n_in, e.g. Llama gate / up 4096 -> 14336): the feature
    covariance W Ex W^T has rank <= n_in, so its leading eigenvectors come from an n_in-sized problem
    (ptd_eigh_factored).  PTD_INPUT_COVARIANCE=0 disables the route, =1 forces it when legal."""
    import os

    if top_k is None or top_k > n_in or n_out <= n_in:
        return False
    flag = os.environ.get("PTD_INPUT_COVARIANCE", "auto")
    if flag == "0":
        return False
    return True if flag == "1" else n_out >= n_in + n_in // 2


class StepBatch:
    """The bf16 activation matrices of up to PTD_SYRK_STEPS (default 8) calibration steps, held back so that their
    covariance sums reach the f64 accumulator in ONE pass (ops.syrk_accumulate_multi): dwain.py:147-152 runs once per
    step and every call reads and writes the live triangle of E -- 134 MB at n = 4096 against 17 MB of activations at
    2048 tokens -- which, not the matrix cores, bounds it.  A matrix the caller may still write to
    (a view of a module's input or output) is copied into the batch; one nobody else holds (the product the stand-in
    formed itself) is kept by reference.  Everything held is added before E is read (`flush`).  The bytes held by all
    batches of the process are bounded (PTD_SYRK_BUFFER_MB, default 4096): beyond that a step is added at once, as in
    round 4.  f32 activations are never held (their product is bound by the matrix cores)."""

    held_bytes = 0           # over every batch of the process
    import threading as _threading
    _lock = _threading.Lock()

    def __init__(self, E: torch.Tensor):
        import os

        import weakref

        self.E = E
        self.max_steps = max(1, int(os.environ.get("PTD_SYRK_STEPS", "8")))
        self.budget = int(os.environ.get("PTD_SYRK_BUFFER_MB", "4096")) << 20
        self.pending: list = []
        # a batch dropped with steps still held (an exception during calibration, a layer skipped before its
        # eigenvectors) gives its bytes back: the counter is process-wide and would otherwise stay raised for good,
        # until every later batch fell back to one call per step
        self._held = [0]
        weakref.finalize(self, StepBatch._release, self._held)

    @staticmethod
    def _release(held: list) -> None:
        with StepBatch._lock:
            StepBatch.held_bytes -= held[0]
            held[0] = 0

    @staticmethod
    def holdable(y: torch.Tensor) -> bool:
        """bf16 matrices on the device: their covariance product is bound by the accumulator's traffic (f32 ones by
        the matrix cores: nothing to gain from holding them)."""
        return y.dtype == torch.bfloat16 and y.is_cuda and y.dim() == 2

    def add(self, y: torch.Tensor, private: bool = False) -> None:
        nbytes = y.numel() * y.element_size()
        if self.max_steps == 1 or not self.holdable(y) or StepBatch.held_bytes + nbytes > self.budget:
            self.flush()
            ops.syrk_accumulate(self.E, y, 1.0 / y.shape[0])
            return
        if self.pending and (self.pending[0].shape != y.shape):
            self.flush()
        self.pending.append(y if private else y.clone(memory_format=torch.contiguous_format))
        with StepBatch._lock:
            StepBatch.held_bytes += nbytes
            self._held[0] += nbytes
        if len(self.pending) >= self.max_steps:
            self.flush()

    def flush(self) -> None:
        if not self.pending:
            return
        ys, self.pending = self.pending, []
        with StepBatch._lock:
            StepBatch.held_bytes -= self._held[0]
            self._held[0] = 0
        ops.syrk_accumulate_multi(self.E, ys, 1.0 / ys[0].shape[0])
        # (a flush from a worker thread of run_concurrently runs on that thread's side stream while the matrices were
        # allocated on the caller's: the allocator must not hand their memory out again before this stream is done
        # with it.  The drivers flush on the caller's stream before the concurrent section -- SharedInputPool.finalize,
        # the reductions -- so this is the safety net, not the rule.)
        if self.E.is_cuda:
            cur = torch.cuda.current_stream(self.E.device)
            for y in ys:
                y.record_stream(cur)


class Covariance:
    """sum over calibration steps of y^T y / T (and of mean_rows(y)) for one layer, in HBM.

    With ``weight`` and ``top_k`` given and n_out sufficiently larger than n_in the INPUT second
    moment x^T x / T is accumulated instead (n_in^2 instead of n_out^2 entries, no layer-output GEMM
    needed) and the eigenvectors of W Ex W^T -- the same matrix -- are obtained through
    ``ops.eigh_factored``."""

    def __init__(self, n: int, device: torch.device, float64: bool, with_mean: bool = False,
                 weight: Optional[torch.Tensor] = None, top_k: Optional[int] = None):
        dt = torch.float64 if float64 else torch.float32
        self.weight = weight
        self.input_route = weight is not None and _input_route_wanted(n, weight.shape[1], top_k)
        m = weight.shape[1] if self.input_route else n
        self.E = torch.zeros((m, m), dtype=dt, device=device)
        # mean of the accumulated rows (falor): of y, or of x on the input route (Ey = W mean(x))
        self.ey = torch.zeros(m, dtype=dt, device=device) if with_mean else None
        self.steps = 0
        self.batch = StepBatch(self.E)

    def add_features(self, y: torch.Tensor, private: bool = False) -> None:
        """y: [T, n] layer output rows (dwain.py:147-152; falor.py:160-161).  `private`: nobody else holds or will
        write to `y` (it may be kept by reference until the batch of steps is added)."""
        assert not self.input_route
        t = y.shape[0]
        self.batch.add(y, private)
        if self.ey is not None:
            ops.colsum_accumulate(self.ey, y, 1.0 / t)
        self.steps += 1

    def add_inputs(self, x_rows: torch.Tensor, weight2d: torch.Tensor, features: Optional[torch.Tensor] = None):
        """x_rows [T, n_in], weight2d [n, n_in].  Direct route: y = x W^T on the matrix cores (or the
        caller's `features` if it already has them), then Y^T Y.  Input route: X^T X only."""
        if self.input_route:
            self.batch.add(x_rows)
            if self.ey is not None:
                ops.colsum_accumulate(self.ey, x_rows, 1.0 / x_rows.shape[0])
            self.steps += 1
            return
        if features is not None:
            self.add_features(features)
        else:
            self.add_features(ops.matmul(x_rows, weight2d.T), private=True)

    def reduce_to_owner(self, shard, index: int) -> None:
        """Sum the partial statistics of all ranks on the owner of `index` (the one bulk exchange of the path:
        RCCL over xGMI; packed lower triangle, see sharding.py).  Only the owner may call eigenvectors()."""
        self.batch.flush()
        with phase("comm"):
            shard.reduce_lower_to_owner(self.E, index)
        small = [torch.tensor([float(self.steps)], dtype=torch.float64, device=self.E.device)]
        if self.ey is not None:
            small.append(self.ey.double())
        flat = torch.cat(small)
        shard.reduce_small_to_owner(flat, index)
        if shard.owns(index):
            self.steps = int(round(flat[0].item()))
            if self.ey is not None:
                self.ey.copy_(flat[1:].to(self.ey.dtype))

    def reduce_to_owner_async(self, shard, index: int):
        """reduce_to_owner without waiting: the exchange is started (every rank, same order) and the returned
        function completes it -- the owner calls it right before eigenvectors(), on the thread / stream that runs
        the eigensolver, so the sums of the other layers of the pass travel while it works."""
        self.batch.flush()
        with phase("comm"):
            done_e = shard.reduce_lower_to_owner_async(self.E, index)
        small = [torch.tensor([float(self.steps)], dtype=torch.float64, device=self.E.device)]
        if self.ey is not None:
            small.append(self.ey.double())
        flat = torch.cat(small)
        done_s = shard.reduce_small_to_owner_async(flat, index)

        def complete() -> None:
            done_e()
            done_s()
            if shard.owns(index):
                self.steps = int(round(flat[0].item()))
                if self.ey is not None:
                    self.ey.copy_(flat[1:].to(self.ey.dtype))
        return complete

    def eigen_order(self, top_k: Optional[int]) -> tuple:
        """(order, k) of the eigenproblem `problem` will pose -- known without touching the device, so that the
        problems of a precompute pass can be grouped before any of them is formed."""
        if self.input_route:
            n = (self.weight.shape[1] + 63) // 64 * 64       # (the factored problem is padded to whole 64 x 64 tiles)
            return n, max(1, min(int(top_k), self.weight.shape[1]))
        n = self.E.shape[0]
        return n, (n if top_k is None else max(1, min(int(top_k), n)))

    def problem(self, damp_factor: float, use_mean: bool = False, top_k: Optional[int] = None) -> "EighProblem":
        """Finalise (divide by steps, optional mean removal, Tikhonov damping) and pose the eigenproblem whose solution
        gives the layer's eigenvectors (dwain.py:155-163, falor.py:192-208)."""
        self.batch.flush()
        if self.input_route:
            # cov = W (Ex - mx mx^T) W^T when the mean is removed (falor.py:196-199); damping only shifts eigenvalues
            ex = ops.cov_finalize(self.E, self.steps, 0.0, self.ey if use_mean else None)
            return problem_from_input_moment(self.weight, ex, damp_factor, top_k, factored=True)
        c = ops.cov_finalize(self.E, self.steps, damp_factor, self.ey if use_mean else None)
        return EighProblem(c, top_k)

    def eigenvectors(self, damp_factor: float, use_mean: bool = False, top_k: Optional[int] = None) -> torch.Tensor:
        """The eigenvectors in columns, ascending, f64.  With ``top_k`` only the last top_k columns (largest
        eigenvalues) are formed: [n, top_k]."""
        return self.problem(damp_factor, use_mean, top_k).solve()


class EighProblem:
    """One symmetric eigenproblem of the path: `matrix` [n, n] f64 (finalised, damped), the k largest eigenpairs wanted,
    and `finish(w, v)` that turns them into the layer's eigenvectors (identity for a feature covariance; u = W L^-T s for
    the factored problem of a widening layer).  Posed by Covariance / MomentCovariance, solved alone (`solve`) or together
    with others of the same order (`solve_eigenproblems`)."""

    def __init__(self, matrix: torch.Tensor, k: Optional[int], finish=None):
        n = matrix.shape[0]
        self.matrix = matrix
        self.k = n if k is None else max(1, min(int(k), n))
        self.finish = finish if finish is not None else (lambda w, v: v)

    @property
    def key(self) -> tuple:
        return (self.matrix.shape[0], self.k, self.matrix.stride(0))

    def solve(self) -> torch.Tensor:
        w, v = ops.eigh(self.matrix, self.k, all_values=False)
        return self.finish(w, v)


def problem_from_input_moment(weight: torch.Tensor, ex: torch.Tensor, damp_factor: float, top_k: Optional[int],
                              factored: bool) -> EighProblem:
    """The eigenproblem behind the eigenvectors of the feature covariance C = W Ex W^T of y = x W^T, given the
    finalised INPUT second moment Ex [n_in, n_in] (f64, full symmetric).  ``factored``: the n_in-sized
    problem of ``ops.eigh_factored`` (legal for top_k <= n_in); otherwise, or when W^T W is not safely positive
    definite, C is formed explicitly by two f64 MFMA products."""
    w2d = weight if weight.dim() == 2 else weight[..., 0, 0]
    if factored and top_k is not None and top_k <= w2d.shape[1]:
        fp = ops.eigh_factored_prepare(w2d, ex, top_k)
        if fp is not None:
            return EighProblem(fp.matrix, fp.k, finish=lambda w, v: fp.finish(w, v)[1])
        warn_once("factored-refused", "ptdeco_amd: W^T W of a widening layer is not safely positive definite; "
                  "its feature covariance W Ex W^T is decomposed directly (n_out-sized eigenproblem)")
    w64 = w2d.double()
    # ptd_cov_finalize mirrors the lower triangle (exact symmetry) and adds the damping
    c = ops.cov_finalize(ops.matmul(ops.matmul(w64, ex), w64.T), 1, damp_factor)
    return EighProblem(c, top_k)


def eigenvectors_from_input_moment(weight: torch.Tensor, ex: torch.Tensor, damp_factor: float, top_k: Optional[int],
                                   factored: bool) -> torch.Tensor:
    return problem_from_input_moment(weight, ex, damp_factor, top_k, factored).solve()


def solve_eigenproblems(posers: list, orders: list, device: torch.device, costs: Optional[list] = None) -> list:
    """The eigendecompositions of one precompute pass (dwain.py:580-633: a loop of get_eigenvectors calls).
    posers[i]() -> EighProblem, orders[i] = (n, k) it will have, costs[i] its relative cost (eigh_cost_hint).  The result
    list is in the order of `posers`.

    Work units: problems of one (n, k) that the direct reduction serves are formed and solved PTD_EIGH_BATCH_MAX (default
    2) at a time by ONE ops.eigh_batched call -- their reductions advance in lockstep, every launch serves all of them --
    and finished before the next chunk is formed (few matrices and factored workspaces alive at once); every other
    problem (the filtered route's requests, lone orders) is a unit of its own.
    Lanes: PTD_EIGH_LANES (default 3) fixed lists of units, each run in order on its own stream by its own host thread.
    The units are dealt longest first to the lane with the least work so far (costs from the orders alone), so a lane
    of batched direct reductions -- chains of short launches bound by latency and by the stream of the trailing triangle
    from HBM -- runs beside the filtered route's f64 products on the matrix cores, and two batched chains beside each
    other interleave their columns.  Measured on one Llama-3-8B-width block (three (4096, 2048) direct problems, two
    filtered (4096, 1024), two (1024, 512); B_eigh, gpurun_out/r06_block_c.txt): everything on one stream 238 ms; two
    lanes with batches of <= 2 / 3 / 4: 247 / 196 / 196; three lanes with batches of <= 2: 193.5 +- 0.2 (two blocks:
    400.6 +- 0.1; with the filtered problems costed like direct ones, eigh_cost_hint: 184 / 389); round 5's seven chains
    dealt dynamically to four threads 184-190 (two blocks 397-404).  The three
    direct problems stream 275 GB of trailing triangles whatever the arrangement (~57 ms at the rate the SYMV reaches),
    and every arrangement lands within 5 % of the others once three chains share the chip: what the lanes buy is that
    the time no longer depends on which hardware queue a stream happens to sit on or on which chain finishes first.
    Which unit runs in which lane, and in which order, is decided before anything runs -- not by which chain happens to
    finish first --, and the eigensolver's route memory is per host thread (ptd_eigh_forget_declines), so the results
    do not depend on scheduling.  PTD_EIGH_LANES=1: every unit on the caller's stream from the caller's thread."""
    import os

    from . import _hip

    nlanes = max(1, int(os.environ.get("PTD_EIGH_LANES", "3")))
    lib = _hip.load() if device.type == "cuda" else None
    if costs is None:
        costs = [float(n) ** 3 for n, _k in orders]
    groups: dict = {}
    for i, key in enumerate(orders):
        groups.setdefault(tuple(key), []).append(i)
    out: list = [None] * len(posers)
    routes = {key: (int(lib.ptd_eigh_route(key[0], key[1], 0)) if lib is not None else 1) for key in groups}
    # A group the filtered route would serve joins the direct batches when the pass ALREADY runs direct reductions of its
    # order in other lanes (and there are lanes): beside them the filter's f64 products only share the matrix cores --
    # q / o of two Llama blocks: four filtered problems of 75-127 ms each in the pass (one of them declining late into a
    # 141-ms reduction of its own) against one more batch of four at ~40 ms a matrix; B_eigh 389 -> 338 ms.  Alone (one
    # layer, PTD_EIGH_LANES=1, a pass of filtered problems only) the filtered route keeps its 25 ms against 45.
    # PTD_EIGH_DIRECT_IN_PASS=0: never.
    direct_orders = {key[0] for key, r in routes.items() if r == 1 and len(groups[key]) >= 1 and key[0] >= 2048}
    forced: set = set()
    if nlanes > 1 and device.type == "cuda" and os.environ.get("PTD_EIGH_DIRECT_IN_PASS", "1") != "0":
        forced = {key for key, r in routes.items() if r == 3 and len(groups[key]) >= 2 and key[0] in direct_orders}
    # matrices per launch: PTD_EIGH_BATCH_MAX, by default what fills the lanes evenly -- the large direct problems of the
    # pass over the lanes, between 2 and 4 (one Llama block: 5 problems of order 4096, batches of <= 2, 184 ms; two
    # blocks: 10, batches of <= 4, 338 ms against 357 with pairs)
    if os.environ.get("PTD_EIGH_BATCH_MAX"):
        cap = max(1, int(os.environ["PTD_EIGH_BATCH_MAX"]))
    else:
        big = sum(len(groups[key]) for key, r in routes.items() if key[0] >= 2048 and (r == 1 or key in forced))
        cap = min(4, max(2, -(-big // nlanes)))

    def batch_unit(chunk, direct=False):
        def run():
            problems = [posers[i]() for i in chunk]
            # (a poser may fall back to another order -- a refused factored problem: those are solved alone)
            same = [j for j, p in enumerate(problems) if p.key == problems[0].key]
            if len(same) >= 2:
                pairs = ops.eigh_batched([problems[j].matrix for j in same], problems[0].k, all_values=False,
                                         direct=direct)
                for j, (w, v) in zip(same, pairs):
                    out[chunk[j]] = problems[j].finish(w, v)
            else:
                same = []
            for j, p in enumerate(problems):
                if j not in same:
                    out[chunk[j]] = p.solve()
        return run

    def single_unit(i):
        def run():
            out[i] = posers[i]().solve()
        return run

    timeline = os.environ.get("PTD_EIGH_TIMELINE") == "1" and device.type == "cuda"
    marks: list = []        # (label, start event, end event): PTD_EIGH_TIMELINE=1, read back through LAST_TIMELINE

    def timed(label, run):
        if not timeline:
            return run

        def wrapped():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run()
            e1.record()
            marks.append((label, e0, e1))
        return wrapped

    units: list = []        # (cost, first member, run, matrix-core-bound?)
    for (n, k), members in groups.items():
        route = 1 if (n, k) in forced else routes[(n, k)]
        if route == 1 and len(members) >= 2 and cap >= 2:
            # chunks of `cap`; a lone last member joins the chunk before it (three matrices: one batch of three at cap 2)
            chunks = [members[c0:c0 + cap] for c0 in range(0, len(members), cap)]
            if len(chunks) >= 2 and len(chunks[-1]) == 1 and cap >= 3:
                chunks[-2].extend(chunks.pop())
            for chunk in chunks:
                if len(chunk) >= 2:
                    units.append((0.85 * sum(costs[i] for i in chunk), chunk[0],
                                  timed(f"batch{len(chunk)} n={n} k={k}", batch_unit(chunk, (n, k) in forced)), False))
                else:
                    units.append((costs[chunk[0]], chunk[0], timed(f"single n={n} k={k}", single_unit(chunk[0])), False))
        else:
            units.extend((costs[i], i, timed(f"single n={n} k={k} route={route}", single_unit(i)), route == 3) for i in members)
    # longest first (ties: model order) to the lane with the least work so far; inside a lane in dealing order
    units.sort(key=lambda u: (-u[0], u[1]))
    if device.type != "cuda":
        nlanes = 1
    lanes: list = [[] for _ in range(min(nlanes, max(len(units), 1)))]
    load = [0.0] * len(lanes)
    for cost, _first, run, _heavy in units:
        j = min(range(len(lanes)), key=lambda q: (load[q], q))
        lanes[j].append(run)
        load[j] += cost
        if timeline:
            marks.append((f"plan lane {j} cost {cost:.3g}", None, None))

    def lane(runs):
        def run():
            for r in runs:
                r()
            return None
        return run

    t_begin = None
    if timeline:
        t_begin = torch.cuda.Event(enable_timing=True)
        t_begin.record()
    if len(lanes) == 1:
        lane(lanes[0])()
    else:
        run_concurrently([lane(r) for r in lanes], device, max_streams=len(lanes))
        cur = torch.cuda.current_stream(device)
        for t in out:       # (allocated on a lane's stream, used from here on under the caller's)
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(cur)
    if timeline:
        global LAST_TIMELINE
        torch.cuda.synchronize(device)
        LAST_TIMELINE = [label for label, e0, _e1 in marks if e0 is None] + \
            sorted((round(t_begin.elapsed_time(e0), 1), round(t_begin.elapsed_time(e1), 1), label)
                   for label, e0, e1 in marks if e0 is not None)
    return out


LAST_TIMELINE: list = []


class InputMoment:
    """sum over calibration steps of x^T x / T for ONE input tensor that several layers read (q/k/v, gate/up of
    a transformer block): accumulated once per step, whoever of the sharing layers runs first."""

    def __init__(self, n_in: int, device: torch.device, float64: bool):
        self.E = torch.zeros((n_in, n_in), dtype=torch.float64 if float64 else torch.float32, device=device)
        self.steps = 0
        self.ex: Optional[torch.Tensor] = None
        self.step_id = -1     # the pool's step in which the moment was last added to
        self.step_key = None  # identity of that step's input tensor
        self.batch = StepBatch(self.E)

    def add(self, x_rows: torch.Tensor) -> None:
        self.batch.add(x_rows)
        self.steps += 1

    def all_reduce(self, shard) -> None:
        """A shared moment feeds layers with different owners: its lower triangle is summed on every rank."""
        self.batch.flush()
        shard.all_reduce_lower(self.E)
        steps = torch.tensor([self.steps], dtype=torch.int64, device=self.E.device)
        shard.all_reduce_small(steps)
        self.steps = int(steps.item())

    def finalize(self) -> None:
        self.batch.flush()
        self.ex = ops.cov_finalize(self.E, self.steps, 0.0)


class MomentCovariance:
    """A layer's feature covariance expressed through a (shared) input moment: C = W Ex W^T."""

    def __init__(self, weight: torch.Tensor, moment: InputMoment, factored: bool):
        self.weight, self.moment, self.factored = weight, moment, factored

    def eigen_order(self, top_k: Optional[int]) -> tuple:
        n_out, n_in = self.weight.shape[0], self.weight.shape[1]
        if self.factored and top_k is not None and top_k <= n_in:
            return (n_in + 63) // 64 * 64, max(1, min(int(top_k), n_in))
        return n_out, (n_out if top_k is None else max(1, min(int(top_k), n_out)))

    def problem(self, damp_factor: float, use_mean: bool = False, top_k: Optional[int] = None) -> EighProblem:
        assert not use_mean and self.moment.ex is not None, "SharedInputPool.finalize() must run first"
        return problem_from_input_moment(self.weight, self.moment.ex, damp_factor, top_k, self.factored)

    def eigenvectors(self, damp_factor: float, use_mean: bool = False, top_k: Optional[int] = None) -> torch.Tensor:
        return self.problem(damp_factor, use_mean, top_k).solve()


def _tensor_key(t: torch.Tensor):
    """Identity of an input tensor within one forward; None when it cannot be told (inference tensors carry no
    version counter: such a layer then keeps its own statistics, like mode "off")."""
    try:
        version = t._version
    except RuntimeError:
        return None
    return (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype, version)


class SharedInputPool:
    """Covariance statistics of all stand-in layers of one precompute pass (dwain.py:580-633), with ONE x^T x
    per distinct input tensor.

    Sigma_y = W Sigma_x W^T, so layers that read the same tensor (q / k / v, gate / up) can share one input
    moment; a layer then gets its eigenvectors either through the n_in-sized factored problem (widening layers)
    or from the explicit product W Ex W^T.  Which layers share is discovered on the first forward: every stand-in
    reports (its input rows, its output rows); identical tensors (same storage, shape, strides, version -- the
    references are held until the step ends, so an address cannot be reused meanwhile) form a group.

    ``mode`` (PTD_SHARE_INPUT_COVARIANCE): "off" = every layer on its own (widening layers still use their own
    x^T x); "all" = every group of >= 2 layers shares; "auto" (default) = widening layers of a group always share,
    the others only when the f64 products W Ex W^T cost less than their D output-side SYRKs (they do for the
    D >= ~32 calibration steps of real runs; with a handful of steps y^T y is cheaper)."""

    # Tunables of the "auto" decision only (which of two equivalent routes is cheaper; results agree to f64
    # rounding either way): MEASURED rates on MI355X of the covariance SYRK with an f64 accumulator (on the triangle's
    # T n (n + 1) flops: f32 activations 104-109 TFLOP/s, bf16 490-620 at the calibration shapes -- bench.py `kernels`,
    # profiles/pmc_syrk_r04.json) and of the f64 products W Ex W^T on the LDS-DMA kernel (55-64 TFLOP/s at these
    # shapes, profiles/f64_gemm_probe_r03.json).  PTD_SHARE_RATE_SYRK_TFLOPS / PTD_SHARE_RATE_F64_TFLOPS override.
    RATE_F32 = 105e12
    RATE_BF16 = 550e12
    RATE_F64 = 55e12

    def __init__(self, num_data_steps: int, float64: bool, device: torch.device, mode: Optional[str] = None):
        import os

        self.mode = (mode or os.environ.get("PTD_SHARE_INPUT_COVARIANCE", "auto")).lower()
        if self.mode not in ("off", "all", "auto"):
            raise ValueError(f"PTD_SHARE_INPUT_COVARIANCE={self.mode!r}: expected off, all or auto")
        self.num_data_steps, self.float64, self.device = num_data_steps, float64, device
        forced = os.environ.get("PTD_SHARE_RATE_SYRK_TFLOPS")
        self.rate_syrk = {torch.bfloat16: float(forced) * 1e12 if forced else self.RATE_BF16}
        self.rate_syrk_default = float(forced) * 1e12 if forced else self.RATE_F32
        self.rate_f64 = float(os.environ.get("PTD_SHARE_RATE_F64_TFLOPS", self.RATE_F64 / 1e12)) * 1e12
        self.members: list = []
        self.discovered = False
        self.groups: list[list] = []    # for introspection / tests: lists of member names
        self._pending: list = []
        self._accumulate = True
        self._step = -1

    def register(self, member) -> None:
        """member: has .weight [n_out, n_in], .top_k, .name; receives .cov (Covariance | MomentCovariance)
        and .moment (InputMoment | None)."""
        member.cov, member.moment = None, None
        self.members.append(member)

    def begin_step(self, accumulate: bool = True) -> None:
        self._step += 1
        self._accumulate = accumulate

    def observe(self, member, x_rows: torch.Tensor, y_rows: torch.Tensor) -> None:
        if not self.discovered:
            self._pending.append((member, x_rows, y_rows))  # held until end_step
            return
        if self._accumulate:
            self._add(member, x_rows, y_rows)

    def _add(self, member, x_rows, y_rows) -> None:
        mom = member.moment
        if mom is None:
            member.cov.add_inputs(x_rows, member.weight, features=y_rows)
            return
        key = _tensor_key(x_rows)
        if mom.step_id != self._step:
            mom.step_id, mom.step_key = self._step, key
            mom.add(x_rows)
        elif mom.step_key != key or key is None:
            raise RuntimeError(f"ptdeco_amd: layer {member.name} shared its input with other layers on the first "
                               "calibration step but not on this one; set PTD_SHARE_INPUT_COVARIANCE=off")

    def end_step(self) -> None:
        if self.discovered:
            return
        # a stand-in called more than once per forward (tied layers, a layer reused across time steps) keeps its own
        # statistics and accumulates at every call, as the reference does (dwain.py:166-208); so does a layer whose
        # input tensor has no identity (inference tensors)
        calls: dict = {}
        for member, _, _ in self._pending:
            calls[id(member)] = calls.get(id(member), 0) + 1
        by_key: dict = {}
        alone: list = []
        for member, x_rows, _ in self._pending:
            key = _tensor_key(x_rows)
            if calls[id(member)] > 1 or key is None:
                if all(member is not a for a, _ in alone):
                    alone.append((member, x_rows.shape[0]))
                continue
            by_key.setdefault(key, []).append(member)
        t_rows = {id(m): x.shape[0] for m, x, _ in self._pending}
        for members in by_key.values():
            self._plan_group(members, t_rows[id(members[0])])
        for member, rows in alone:
            self._plan_group([member], rows)
        for m in self.members:
            if m.cov is None:  # not reached by the forward: same error as a tap that saw no input
                raise RuntimeError(f"layer {m.name} was not reached by the model's forward")
        self.discovered = True
        if self._accumulate:
            for member, x_rows, y_rows in self._pending:
                self._add(member, x_rows, y_rows)
        self._pending.clear()

    def _plan_group(self, members: list, t_rows: int) -> None:
        n_in = members[0].weight.shape[1]
        wide = [m for m in members if _input_route_wanted(m.weight.shape[0], n_in, m.top_k)]
        rest = [m for m in members if m not in wide]
        d = self.num_data_steps
        rate = self.rate_syrk.get(members[0].weight.dtype, self.rate_syrk_default)   # (activations come in the weight dtype)
        direct = {id(m): d * t_rows * m.weight.shape[0] ** 2 / rate for m in rest}
        explicit = {id(m): (2.0 * m.weight.shape[0] * n_in * n_in + 2.0 * m.weight.shape[0] ** 2 * n_in) / self.rate_f64
                    for m in rest}
        if self.mode == "off" or len(members) < 2:
            shared_wide, shared_rest = [], []
        elif self.mode == "all":
            shared_wide, shared_rest = wide, rest
        else:
            shared_wide = wide if len(wide) >= 2 or (wide and rest) else []
            gain = [m for m in rest if explicit[id(m)] < direct[id(m)]]
            moment_cost = 0.0 if shared_wide else d * t_rows * n_in * n_in / rate
            saved = sum(direct[id(m)] - explicit[id(m)] for m in gain)
            shared_rest = gain if (gain and saved > moment_cost and (shared_wide or len(gain) >= 2 or wide)) else []
            if shared_rest and not shared_wide:
                shared_wide = wide
        sharing = shared_wide + shared_rest
        if len(sharing) >= 2:
            moment = InputMoment(n_in, self.device, self.float64)
            for m in sharing:
                m.moment = moment
                m.cov = MomentCovariance(m.weight.detach(), moment, factored=m in shared_wide)
            self.groups.append([m.name for m in sharing])
        for m in members:
            if m.cov is None:
                m.cov = Covariance(m.weight.shape[0], self.device, self.float64, weight=m.weight.detach(), top_k=m.top_k)

    def moments(self) -> list:
        out: list = []
        for m in self.members:
            if m.moment is not None and all(m.moment is not o for o in out):
                out.append(m.moment)
        return out

    def reduce(self, shard) -> list:
        """Sum the partial statistics over the ranks: shared moments everywhere (their layers have different
        owners), an unshared layer's statistics on its owner only (member i of the pass is owned by rank i % G).
        The per-layer sums are STARTED here, in member order on every rank, and completed by the functions returned
        (one per member, None where there is nothing to wait for): the owner of member i calls its function right
        before the eigensolve, so layer i + 1's sum is on the wire while layer i is being decomposed."""
        for mom in self.moments():
            mom.all_reduce(shard)
        return [m.cov.reduce_to_owner_async(shard, i) if m.moment is None else None
                for i, m in enumerate(self.members)]

    def finalize(self) -> None:
        """Shared Ex matrices, formed once on the caller's stream before the (concurrent) eigendecompositions; the
        calibration steps the members' own statistics still hold back are added here too, on the caller's stream."""
        for mom in self.moments():
            mom.finalize()
        for m in self.members:
            batch = getattr(m.cov, "batch", None)
            if batch is not None:
                batch.flush()


def eigh_route_hint(cov, n_out: int, top_k: Optional[int]) -> int:
    """Which solver the eigendecomposition behind `cov` (Covariance | MomentCovariance) will be given first
    (ptd_eigh_route): 3 = filtered subspace iteration -- chip-filling f64 products, gains nothing from running beside
    other chains and loses its resident Rayleigh-Ritz kernels when it does --, 1 = the direct reduction, a latency-bound
    chain of short launches that overlaps well with others."""
    from . import _hip

    if isinstance(cov, MomentCovariance):
        n = cov.weight.shape[1] if cov.factored else n_out
    else:
        n = cov.E.shape[0]
    k = n if top_k is None else max(1, min(int(top_k), n))
    return int(_hip.load().ptd_eigh_route(n, k, 0))


def eigh_cost_hint(cov, n_out: int, top_k: Optional[int]) -> float:
    """Relative cost of the eigendecomposition behind `cov`, for the dealing of units to lanes (longest first): the order
    of the eigenproblem cubed, the factored route (Llama gate / up) with its products over the n_out rows on top.  A
    filtered-route problem counts like a direct one of its order (PTD_FILTERED_COST, default 1.0): alone it takes half
    the time, but in a pass it runs beside other lanes' work and its f64 products share the matrix cores -- 65-117 ms
    against 210-236 for a batch of two direct reductions (PTD_EIGH_TIMELINE=1); at round 5's 0.45 the four filtered
    problems of two Llama blocks were dealt two and two to the lanes whose batches end first and the third lane idled
    for 100 ms (B_eigh 403 ms; 389 with 1.0, one block 195 -> 184)."""
    n = cov.weight.shape[1] if isinstance(cov, MomentCovariance) and cov.factored else \
        (n_out if isinstance(cov, MomentCovariance) else cov.E.shape[0])
    route = eigh_route_hint(cov, n_out, top_k)
    import os

    cost = float(n) ** 3 * (float(os.environ.get("PTD_FILTERED_COST", "1.0")) if route == 3 else 1.0)
    if isinstance(cov, MomentCovariance) and cov.factored:
        cost += 0.25 * float(n_out) * float(n) ** 2
    return cost


_DEDICATED_STREAMS: dict = {}  # (device index, CU split) -> streams with hardware queues of their own


def dedicated_streams(device: torch.device, want: int) -> list:
    """`want` HIP streams of `device` that each own a hardware queue (ptd_stream_create_dedicated:
    hipExtStreamCreateWithCUMask), created once per device and kept for the life of the process.  Round 5 took streams
    from torch's pool and MEASURED which of them shared one of the runtime's four hardware queues (chain_streams below,
    still there behind PTD_LANE_STREAMS=pool): two chains on one queue run packet by packet, and which streams share
    depends on the creation order of every stream in the process.  A stream created with a CU mask is given a queue no
    other stream uses: the lanes overlap by construction, nothing to probe, nothing that can change mid-process.
    PTD_LANE_CUS = "a,b,c" gives lane i a contiguous range of that many CUs (experiments); default: every lane sees the
    whole chip."""
    import ctypes
    import os
    import threading

    from . import _hip

    global _CHAIN_STREAMS_LOCK
    if _CHAIN_STREAMS_LOCK is None:
        _CHAIN_STREAMS_LOCK = threading.Lock()
    index = device.index if device.index is not None else torch.cuda.current_device()
    split = os.environ.get("PTD_LANE_CUS", "")
    counts = [int(c) for c in split.split(",") if c.strip()] if split else []
    with _CHAIN_STREAMS_LOCK:
        have = _DEDICATED_STREAMS.setdefault((index, split), [])
        lib = _hip.load()
        with torch.cuda.device(index):
            while len(have) < want:
                i = len(have)
                first = sum(counts[:i]) if i < len(counts) else 0
                count = counts[i] if i < len(counts) else 0
                ptr = ctypes.c_void_p(0)
                _hip.check(lib.ptd_stream_create_dedicated(first, count, ctypes.byref(ptr)), "ptd_stream_create_dedicated")
                if not _DEDICATED_RAW:
                    import atexit

                    atexit.register(_destroy_dedicated_streams)
                _DEDICATED_RAW.append((index, ptr.value))
                have.append(torch.cuda.ExternalStream(ptr.value, device=torch.device("cuda", index)))
        return list(have[:want])


_DEDICATED_RAW: list = []      # (device index, hipStream_t) of every dedicated stream: destroyed at interpreter exit


def _destroy_dedicated_streams() -> None:
    """The streams are the library's own (torch only wraps them): left to the runtime's teardown they outlive a
    profiler's tool (rocprofv3 aborted in its finalisation, round 6); destroyed here, in order, while everything lives."""
    from . import _hip

    try:
        lib = _hip.load()
        for index, raw in _DEDICATED_RAW:
            with torch.cuda.device(index):
                torch.cuda.synchronize(index)
                lib.ptd_stream_destroy(raw)
    except Exception:  # noqa: BLE001  (exit path: nothing to report to)
        pass
    _DEDICATED_RAW.clear()
    _DEDICATED_STREAMS.clear()


_CHAIN_STREAMS: dict = {}      # device index -> (streams on pairwise distinct hardware queues, candidates exhausted?)
CHAIN_STREAM_STATS = {"calls": 0, "selections": 0, "rechecks_failed": 0}     # (bench.py reports them)
_CHAIN_STREAMS_LOCK = None


def chain_streams(device: torch.device, want: int) -> list:
    """Up to `want` HIP streams of `device` that sit on pairwise DIFFERENT hardware queues, for chains of dependent
    launches that are meant to interleave on the GPU (run_concurrently).

    The ROCm runtime multiplexes the streams of a process onto 4 hardware queues per priority level
    (tools/probes/launch_rate_probe.hip: of eight streams created in a row, 0/7, 1/6, 2/5 and 3/4 share a queue; the
    high-priority streams have four queues of their own) and two chains on one queue are executed packet by packet, in
    turn.  Which streams share is a fact of the process (creation order of every stream in it, torch's pools included),
    so it is MEASURED, once per device: candidates from torch's high-priority pool, then from the normal one, are tested
    pairwise with ptd_stream_pair_wall_us (two single-wave kernels that hold their queue for 400 us: side by side
    ~0.42 ms, serialised ~0.82 ms) and taken greedily while they overlap with every stream taken before.  The streams
    are kept for the life of the process (the mapping of an existing stream does not change: the probe's matrix is
    identical before and after use).  Fewer than `want` distinct queues -> fewer streams are returned.
    PTD_CHAIN_STREAMS_VERIFY=0 skips the measurement (pool streams as they come, the round-4 behaviour)."""
    import os
    import threading
    import ctypes

    from . import _hip

    global _CHAIN_STREAMS_LOCK
    if _CHAIN_STREAMS_LOCK is None:
        _CHAIN_STREAMS_LOCK = threading.Lock()
    device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    device = torch.device("cuda", index)
    want = max(1, int(want))
    if os.environ.get("PTD_LANE_STREAMS", "dedicated") != "pool":
        return dedicated_streams(device, want)
    if os.environ.get("PTD_CHAIN_STREAMS_VERIFY", "1") == "0":
        return [torch.cuda.Stream(device=device) for _ in range(want)]
    with _CHAIN_STREAMS_LOCK:
        have, exhausted = _CHAIN_STREAMS.get(index, ([], False))
        lib = _hip.load()
        # (400 us per kernel: a serialised pair shows as +400 us, far above what a descheduled host thread or a slow
        # launch adds -- at 150 us a 65-us hiccup of the host read as "serialised" and set off a re-selection: bench.py on a
        # busy box, bf16 stack 388 -> 509 ms per step)
        spin_us = 400
        wall = ctypes.c_double(0.0)
        CHAIN_STREAM_STATS["calls"] += 1
        if have and (len(have) >= want or exhausted):
            # Re-check the kept streams, all at once (one 150-us kernel on each: 0.2 ms): the mapping of streams onto
            # hardware queues was seen to change within a process -- bench.py: four streams verified distinct at the first
            # call, two of them serialised a minute later (B_eigh 243 instead of 187 ms, one chain ending at 390 ms) --
            # so a verdict is only good for the call it was measured in.
            group = have[:want]
            arr = (ctypes.c_void_p * len(group))(*[h.cuda_stream for h in group])
            with torch.cuda.device(device):
                for attempt in range(2):       # (a failed check is repeated once before it counts)
                    _hip.check(lib.ptd_streams_wall_us(arr, len(group), spin_us, ctypes.byref(wall)), "ptd_streams_wall_us")
                    if wall.value < 1.5 * spin_us:
                        return list(group)
            CHAIN_STREAM_STATS["rechecks_failed"] += 1
            import logging

            logging.getLogger(__name__).info("cuda:%d: the kept streams no longer overlap (%.0f us for %d x %d us): "
                                             "choosing again", index, wall.value, len(group), spin_us)
            have, exhausted = [], False

        def overlap(a, b) -> bool:
            _hip.check(lib.ptd_stream_pair_wall_us(a.cuda_stream, b.cuda_stream, spin_us, ctypes.byref(wall)),
                       "ptd_stream_pair_wall_us")
            return wall.value < 1.5 * spin_us

        with torch.cuda.device(device):
            tried = 0
            # (normal priority only.  The high-priority pool has four hardware queues of its own, but chains on
            # high-priority streams were measured SLOWER inside bench.py's process -- B_eigh 243 ms against 187 -- although
            # the streams passed this very test; on normal-priority streams the test's verdict held)
            prios = [int(p_) for p_ in os.environ.get("PTD_CHAIN_STREAM_PRIORITIES", "0").split(",")]
            for priority in prios:
                for _ in range(16):
                    if len(have) >= want:
                        break
                    cand = torch.cuda.Stream(device=device, priority=priority)
                    tried += 1
                    if any(cand.cuda_stream == h.cuda_stream for h in have):
                        continue
                    if all(overlap(h, cand) for h in have):
                        have.append(cand)
            exhausted = len(have) < want
        _CHAIN_STREAMS[index] = (have, exhausted)
        CHAIN_STREAM_STATS["selections"] += 1
        _log_chain_streams(index, have, tried)
        return list(have[:want])


def _log_chain_streams(index: int, have: list, tried: int) -> None:
    import logging

    logging.getLogger(__name__).info("cuda:%d: %d streams on distinct hardware queues out of %d candidates: %s", index,
                                     len(have), tried, " ".join(hex(h.cuda_stream) for h in have))


def run_concurrently(jobs, device: torch.device, max_streams: Optional[int] = None, routes: Optional[list] = None,
                     costs: Optional[list] = None) -> list:
    """Run independent device-side jobs (callables returning tensors) from separate host threads, each on
    its own HIP stream, and return their results in order.

    One eigendecomposition is a chain of ~8000 short dependent launches that leaves most of the GPU
    idle; chains of different layers issued on different streams interleave on the device (two
    n = 4096 matrices: 1.5x the throughput of running them back to back, three: 1.9x; round 4, with the filtered route
    in the mix: the seven layers of a Llama block 339 ms back to back, 250 on three streams, 204 on four, 211 on five).  The C ABI keeps
    no shared mutable state and releases the GIL, so the host side is plain threads.  The streams are the device's
    chain streams (`chain_streams`: measured to sit on distinct hardware queues, re-checked at every call).  Stream order:
    every side stream first waits for the caller's stream (inputs), the caller's stream waits for all
    of them at the end (outputs).  PTD_EIGH_STREAMS overrides the stream count (1 = sequential).  `routes` (one
    ptd_eigh_route value per job): only read with PTD_EIGH_STREAMS_BY_ROUTE=1, see the comment below.  `costs` (one
    relative cost per job, eigh_cost_hint): with PTD_EIGH_LONGEST_FIRST=1 the workers take the jobs longest first
    (opt-in: the seven layers of a Llama block 226 -> 214 ms when two chains share a hardware queue, 188 -> 190 ms
    when they do not)."""
    import os
    import threading

    jobs = list(jobs)
    device = torch.device(device)
    want = int(os.environ.get("PTD_EIGH_STREAMS", "4")) if max_streams is None else max_streams
    if routes is not None and want > 1 and device.type == "cuda" and os.environ.get("PTD_EIGH_STREAMS_BY_ROUTE", "0") == "1":
        # OPT-IN (PTD_EIGH_STREAMS_BY_ROUTE=1): the jobs the filtered route will take run one after the other on the
        # caller's stream (with the whole chip, and with the resident kernels of their inner eigenproblem); only the
        # latency-bound ones share the chip.  Measured and NOT the default (profiles/streams_r04.json): three filtered
        # chains back to back 171 ms, on three streams 137-181; the 2-block Llama stack 668-681 ms with this rule
        # against 583-622 with every chain on its own stream -- the latency-bound phases of a filtered chain (Lanczos,
        # the Cholesky sweeps, the Rayleigh-Ritz eigenproblem) do overlap with another chain's products.
        alone = [i for i, r in enumerate(routes) if r == 3]
        if alone:
            rest = [i for i in range(len(jobs)) if routes[i] != 3]
            out = [None] * len(jobs)
            for i in alone:
                out[i] = jobs[i]()
            for i, res in zip(rest, run_concurrently([jobs[i] for i in rest], device, max_streams,
                                                     costs=None if costs is None else [costs[i] for i in rest])):
                out[i] = res
            return out
    workers = max(1, min(len(jobs), want))
    if workers == 1 or device.type != "cuda":
        return [job() for job in jobs]
    index = device.index if device.index is not None else torch.cuda.current_device()
    device = torch.device("cuda", index)
    main = torch.cuda.current_stream(device)
    # The chains' streams: kept per device and VERIFIED to sit on pairwise different hardware queues (chain_streams).
    # Round 4 took fresh streams from torch's pool at every call and was bimodal -- B_eigh of a Llama block 187 ms or
    # 226-300 ms: whenever two of the four streams shared a hardware queue, two chains of ~8000 dependent launches ran one
    # packet after the other (PTD_EIGH_JOB_LOG=1: `v`, a 10-ms problem, ended at 148 ms beside `o` on the same queue).
    streams = chain_streams(device, workers)
    workers = len(streams)
    if workers == 1:
        return [job() for job in jobs]
    out: list = [None] * len(jobs)
    errors: list = []
    lock = threading.Lock()
    cursor = [0]
    order = list(range(len(jobs)))
    joblog = [] if os.environ.get("PTD_EIGH_JOB_LOG") else None      # (diagnostic: start / end of every job, ms)
    import time as _time

    tstart = _time.perf_counter()
    if costs is not None and len(costs) == len(jobs) and os.environ.get("PTD_EIGH_LONGEST_FIRST", "0") == "1":
        order.sort(key=lambda i: -costs[i])      # (stable: equal costs keep the model's order)

    def worker(w: int) -> None:
        try:
            torch.cuda.set_device(device)
            with torch.no_grad(), torch.cuda.stream(streams[w]):
                while not errors:
                    with lock:
                        c = cursor[0]
                        cursor[0] += 1
                    if c >= len(jobs):
                        break
                    i = order[c]
                    if joblog is None:
                        out[i] = jobs[i]()
                    else:
                        import time as _t

                        t0 = _t.perf_counter()
                        out[i] = jobs[i]()
                        streams[w].synchronize()
                        joblog.append((i, w, round((t0 - tstart) * 1e3, 1), round((_t.perf_counter() - tstart) * 1e3, 1)))
        except BaseException as exc:  # re-raised on the calling thread
            errors.append(exc)

    from . import _hip

    threads = [threading.Thread(target=worker, args=(w,), name=f"ptdeco-eigh-{w}") for w in range(workers)]
    # interleaved chains: no kernel may claim a whole XCD for itself (ptd_set_concurrent_chains; the hint is kept
    # per device and keyed by the calling thread's current device, so it is set with `device` current)
    with torch.cuda.device(device):
        before = _hip.load().ptd_set_concurrent_chains(workers)
    try:
        for th in threads:
            th.start()
        for th in threads:
            th.join()
    finally:
        with torch.cuda.device(device):
            _hip.load().ptd_set_concurrent_chains(before)
    for st in streams:
        main.wait_stream(st)
    for res in out:  # the results were allocated on a side stream and live on under the caller's
        for t in (res if isinstance(res, (tuple, list)) else (res,)):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(main)
    if joblog is not None:
        import sys

        print("[run_concurrently] streams " + " ".join(hex(st.cuda_stream) for st in streams) + " jobs (i, worker, start, end): "
              + str(sorted(joblog)), file=sys.stderr, flush=True)
    if errors:
        raise errors[0]
    return out


def build_factors(weight2d: torch.Tensor, u: torch.Tensor, rank: int, dtype: torch.dtype, dense: bool = True):
    """Top-`rank` eigenvectors -> (uk [n, r], U [n_in, r], W~ [n, n_in] or None) in `dtype`.

    U = W^T uk and W~ = (U uk^T)^T = uk U^T (dwain.py:424-429; falor.py:346-348); W~ is only
    formed when the candidate cannot be evaluated through the pair (`dense`)."""
    n = u.shape[1]
    uk = u[:, n - rank:].to(dtype).contiguous()
    w = weight2d if weight2d.dtype == dtype else weight2d.to(dtype)
    big_u = ops.matmul(w.T, uk)
    w_deco = ops.matmul(uk, big_u.T) if dense else None
    return uk, big_u, w_deco


class FactorBank:
    """The factors of every candidate rank of one layer from ONE product: U_max = W^T u_max for the
    largest rank that can be asked for; a rank-r candidate takes the trailing r columns of u_max and
    U_max (the eigenvectors are sorted by ascending eigenvalue), which are the very dot products
    ``build_factors`` would form for that rank."""

    def __init__(self, weight2d: torch.Tensor, u: torch.Tensor, max_rank: int, dtype: torch.dtype):
        n = u.shape[1]
        self.max_rank = max(1, min(int(max_rank), n))
        self.uk = u[:, n - self.max_rank:].to(dtype).contiguous()
        self.weight = weight2d if weight2d.dtype == dtype else weight2d.to(dtype)
        self.big_u = ops.matmul(self.weight.T, self.uk)

    def get(self, rank: int, dense: bool = False):
        """(uk [n, r], U [n_in, r] (a column slice), W~ or None) like ``build_factors``."""
        if rank > self.max_rank:
            raise ValueError(f"rank {rank} above the bank's {self.max_rank}")
        lo = self.max_rank - rank
        uk = self.uk[:, lo:].contiguous()
        big_u = self.big_u[:, lo:]
        w_deco = ops.matmul(uk, big_u.T) if dense else None
        return uk, big_u, w_deco


def build_pair(layer: torch.nn.Module, big_u: torch.Tensor, uk: torch.Tensor, dtype: Optional[torch.dtype]):
    """The rank-r replacement of `layer`: first weight U^T [r, n_in], second weight uk [n_out, r],
    bias copied (dwain.py:69-85, 121-144).  A 1x1 conv pair takes default stride / padding /
    dilation exactly like the reference."""
    r = uk.shape[1]
    has_bias = layer.bias is not None
    dev = layer.weight.device
    first_w = big_u.T.contiguous()
    if isinstance(layer, torch.nn.Conv2d):
        m1 = torch.nn.Conv2d(layer.in_channels, r, kernel_size=1, bias=False, device=dev)
        m2 = torch.nn.Conv2d(r, layer.out_channels, kernel_size=1, bias=has_bias, device=dev)
        first_w, second_w = first_w[:, :, None, None], uk[:, :, None, None]
    else:
        m1 = torch.nn.Linear(layer.in_features, r, bias=False, device=dev)
        m2 = torch.nn.Linear(r, layer.out_features, bias=has_bias, device=dev)
        second_w = uk
    with torch.no_grad():
        m1.weight.copy_(first_w)
        m2.weight.copy_(second_w)
        if has_bias:
            m2.bias.copy_(layer.bias)
    pair = torch.nn.Sequential(m1, m2)
    if dtype is not None:
        pair.to(dtype)
    return fuse_pair(pair)


def is_num_params_reduced(proportion: float, in_features: int, out_features: int) -> bool:
    """dwain.py:569-577, falor.py:273-281."""
    return (in_features + out_features) * proportion * min(in_features, out_features) < in_features * out_features
