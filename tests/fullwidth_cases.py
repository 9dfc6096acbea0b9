"""Builders of the FULL-WIDTH parity cases of BASELINE configs[2] (C3) and configs[3] (C4), shared by
tests/test_fullwidth_gpu.py and tools/scan_fullwidth_seeds.py (which runs the CPU oracle alone over a few
seeds and reports how far every step of its run stays from the thresholds it is compared with).

Widths are the real ones (ViT-B/16: 768 / 2304 / 3072 / 1000; Llama-3-8B: 4096 / 1024 / 14336); depth, batch
and step counts are cut so that the CPU oracle finishes in about a minute on the GPU box's host cores.
"""

from __future__ import annotations

import itertools

import torch

import toy_models as tm

# ---------------------------------------------------------------- C3: falor on a ViT-B/16-width model
# "head" (768 -> 1000) reads the class token only: 8 rows per step, 40 in all, so 960 of its 1000 covariance
# eigenvalues are one degenerate (damping) value and the eigenvectors a rank-384 candidate takes from that null
# space are whatever basis the eigensolver happens to return -- LAPACK's in the reference, another one here.  The
# reference's own result is arbitrary there; the layer is blacklisted on BOTH sides (falor.py:439-447).
C3_KW = dict(proportion_threshold=0.9, nsr_final_threshold=0.05, kl_final_threshold=0.01, num_data_steps=5,
             num_metric_steps=2, use_float64=True, use_mean=False, use_damping=True,
             blacklisted_module_names=["head"])
C3_SEEDS = (0, 1)       # (model, data): see tools/scan_fullwidth_seeds.py


def c3_case(model_seed: int = C3_SEEDS[0], data_seed: int = C3_SEEDS[1], depth: int = 3, batch: int = 8):
    """timm vit_base_patch16_224 shapes (d = 768, qkv 2304, mlp 3072, head 1000, 197 tokens per image),
    `depth` blocks; the reference's trainer settings (decompose_falor.yaml:18-22: batch 8, D = 5;
    run_decompose_falor.py:92-93: use_mean=False, use_damping=True) with M = 2 and thresholds at which
    random-weight layers do get replaced."""
    model = tm.ViT(depth=depth)
    tm.init_randn(model, model_seed)
    model.eval()
    g = torch.Generator().manual_seed(data_seed)
    pool = [torch.randn(batch, 3, 224, 224, generator=g) for _ in range(12)]   # x ~ N(0, 1)
    return model, pool


# ---------------------------------------------------------------- C4: dwain on one full-width Llama block
C4_KW = dict(num_data_steps=5, num_metric_steps=1, nsr_final_threshold=1.0, min_rank=32, trade_off_factor=20.0,
             reduction_factor=0.5, max_accepted_ppl_diff=0.4, decompose_in_float64=True,
             blacklisted_module_names=["head", "blocks.0.up"],
             precomputing_covariance_num_splits=None)
C4_SEED = 0
D_MODEL, D_KV, D_FF = 4096, 1024, 14336


class LlamaBlock(torch.nn.Module):
    """RMSNorm -> {q, k, v} -> (q + repeat4(k) + repeat4(v)) -> o -> residual;
    RMSNorm -> down(silu(gate) * up) -> residual, at the Llama-3-8B widths."""

    def __init__(self, d=D_MODEL, kv=D_KV, ff=D_FF):
        super().__init__()
        mk = lambda i, o: torch.nn.Linear(i, o, bias=False)  # noqa: E731
        self.q, self.k, self.v, self.o = mk(d, d), mk(d, kv), mk(d, kv), mk(d, d)
        self.gate, self.up, self.down = mk(d, ff), mk(d, ff), mk(ff, d)
        self.rep = d // kv

    @staticmethod
    def norm(x):
        return x * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + 1e-6).to(x.dtype)

    def forward(self, x):
        h = self.norm(x)
        x = x + self.o(self.q(h) + self.k(h).repeat(1, 1, self.rep) + self.v(h).repeat(1, 1, self.rep))
        h = self.norm(x)
        return x + self.down(torch.nn.functional.silu(self.gate(h)) * self.up(h))


class LlamaStack(torch.nn.Module):
    def __init__(self, blocks=1, d=D_MODEL, kv=D_KV, ff=D_FF, vocab=D_MODEL):
        super().__init__()
        self.blocks = torch.nn.ModuleList(LlamaBlock(d, kv, ff) for _ in range(blocks))
        self.head = torch.nn.Linear(d, vocab, bias=False)

    def forward(self, b):
        x = b["x"]
        for blk in self.blocks:
            x = blk(x)
        return self.head(x)


def seq_ce(batch, logits):
    return torch.nn.functional.cross_entropy(logits.float().reshape(-1, logits.shape[-1]),
                                             batch["targets"].reshape(-1), reduction="none")


def c4_case(seed: int = C4_SEED, tokens: int = 1024, n_batches: int = 10):
    """One block at full width (4096 / 1024 / 14336) + the blacklisted head, f32, W ~ N(0, 1/n_in), inputs
    N(0, 1) x a decaying feature scale [1, tokens, 4096] (1024 tokens per step so that the CPU oracle fits the
    test budget; D = 5 steps give more calibration rows than features), targets = argmax of the original logits.  `up`
    is blacklisted ON BOTH SIDES (a 14336^2 eigendecomposition takes the CPU oracle about a minute on the GPU box's 16
    granted CPUs; `gate`, the same shape, IS decomposed: the factored route end to end against the oracle)."""
    g = torch.Generator().manual_seed(seed)
    model = LlamaStack(1)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(torch.randn(p.shape, generator=g) / p.shape[1] ** 0.5)
    scale = torch.logspace(0, -2, D_MODEL)
    xs = [torch.randn(1, tokens, D_MODEL, generator=g) * scale for _ in range(n_batches)]
    with torch.no_grad():
        batches = [{"x": x, "targets": model({"x": x}).argmax(-1)} for x in xs]
    return model, batches


def cycle(seq):
    return itertools.cycle(list(seq))
