"""Small plain-torch models and deterministic data pools shared by the golden
generator (tests/golden/gen_golden.py) and the parity tests.

Weights and data are stored in the fixtures as tensors; these classes only give
them a module structure (names of the decomposable layers are part of the
decompose_config and therefore of the parity contract).
"""

from __future__ import annotations

import itertools

import torch


def _unwrap(d):
    return d["x"] if isinstance(d, dict) else d


class MLP3(torch.nn.Module):
    """64 -> 128 -> 96 -> 10 with ReLU; accepts a tensor (falor) or {"x": ...} (dwain)."""

    def __init__(self, dims=(64, 128, 96, 10), bias=True):
        super().__init__()
        self.fc1 = torch.nn.Linear(dims[0], dims[1], bias=bias)
        self.fc2 = torch.nn.Linear(dims[1], dims[2], bias=bias)
        self.fc3 = torch.nn.Linear(dims[2], dims[3], bias=bias)

    def forward(self, d):
        x = _unwrap(d)
        return self.fc3(torch.relu(self.fc2(torch.relu(self.fc1(x)))))


class ConvNet(torch.nn.Module):
    """3x3 stem, two 1x1 convs (decomposable), global pool, linear head."""

    def __init__(self, c0=8, c1=48, c2=40, classes=10):
        super().__init__()
        self.stem = torch.nn.Conv2d(3, c0, kernel_size=3, padding=1)
        self.pw1 = torch.nn.Conv2d(c0, c1, kernel_size=1)
        self.pw2 = torch.nn.Conv2d(c1, c2, kernel_size=1, bias=False)
        self.head = torch.nn.Linear(c2, classes)

    def forward(self, d):
        x = _unwrap(d)
        x = torch.relu(self.stem(x))
        x = torch.relu(self.pw1(x))
        x = torch.relu(self.pw2(x))
        return self.head(x.mean(dim=(2, 3)))


class OneLinear(torch.nn.Module):
    """The primitive-test network of the reference (tests/test_deco_primitives_dwain.py:35-50)."""

    def __init__(self, n_in, n_out, bias=True):
        super().__init__()
        self.mod = torch.nn.Linear(n_in, n_out, bias=bias)

    def forward(self, d):
        x = d["inp"] if isinstance(d, dict) else d
        return torch.flatten(self.mod(x), start_dim=1)


class OneConv1x1(torch.nn.Module):
    """tests/test_deco_primitives_dwain.py:53-72."""

    def __init__(self, n_in, n_out, bias=True):
        super().__init__()
        self.mod = torch.nn.Conv2d(n_in, n_out, kernel_size=(1, 1), bias=bias)

    def forward(self, d):
        x = d["inp"] if isinstance(d, dict) else d
        return torch.flatten(self.mod(x), start_dim=1)


def cycle_tensors(pool):
    """Endless iterator over a fixed pool of tensors (falor data stream)."""
    return itertools.cycle(list(pool))


def cycle_dicts(pool, targets=None, key="x"):
    """Endless iterator of {"x": batch[, "targets": t]} dicts (dwain data stream)."""
    items = []
    for i, x in enumerate(pool):
        d = {key: x}
        if targets is not None:
            d["targets"] = targets[i]
        items.append(d)
    return itertools.cycle(items)


def ce_loss(batch, logits):
    """Per-sample cross entropy; dwain exponentiates and averages it
    (reference dwain.py:276-277 applies exp().mean() to whatever loss_fn returns)."""
    return torch.nn.functional.cross_entropy(logits, batch["targets"], reduction="none")


def load_state(model: torch.nn.Module, arrays, prefix: str) -> None:
    """Copy fixture arrays named '<prefix><param name>' into the model."""
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(torch.from_numpy(arrays[prefix + name]))


class _ViTAttention(torch.nn.Module):
    def __init__(self, d, heads):
        super().__init__()
        self.heads = heads
        self.qkv = torch.nn.Linear(d, 3 * d)
        self.proj = torch.nn.Linear(d, d)

    def forward(self, x):
        b, t, d = x.shape
        q, k, v = self.qkv(x).reshape(b, t, 3, self.heads, d // self.heads).permute(2, 0, 3, 1, 4)
        a = torch.nn.functional.scaled_dot_product_attention(q, k, v)
        return self.proj(a.transpose(1, 2).reshape(b, t, d))


class _ViTMlp(torch.nn.Module):
    def __init__(self, d, hidden):
        super().__init__()
        self.fc1 = torch.nn.Linear(d, hidden)
        self.fc2 = torch.nn.Linear(hidden, d)

    def forward(self, x):
        return self.fc2(torch.nn.functional.gelu(self.fc1(x)))


class ViTBlock(torch.nn.Module):
    """timm naming: blocks.N.attn.qkv / attn.proj / mlp.fc1 / mlp.fc2"""

    def __init__(self, d, heads, mlp):
        super().__init__()
        self.norm1 = torch.nn.LayerNorm(d)
        self.attn = _ViTAttention(d, heads)
        self.norm2 = torch.nn.LayerNorm(d)
        self.mlp = _ViTMlp(d, mlp)

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class ViT(torch.nn.Module):
    """ViT-B/16 layout (timm vit_base_patch16_224 layer names and shapes at the default arguments):
    a 16x16 patch convolution (not decomposable), class token, `depth` pre-norm blocks with
    qkv / proj / fc1 / fc2 Linear layers, final norm, classification head."""

    def __init__(self, img=224, patch=16, d=768, depth=12, heads=12, mlp=3072, classes=1000):
        super().__init__()
        self.patch_embed = torch.nn.Conv2d(3, d, kernel_size=patch, stride=patch)
        self.cls_token = torch.nn.Parameter(torch.zeros(1, 1, d))
        self.pos_embed = torch.nn.Parameter(torch.zeros(1, (img // patch) ** 2 + 1, d))
        self.blocks = torch.nn.ModuleList(ViTBlock(d, heads, mlp) for _ in range(depth))
        self.norm = torch.nn.LayerNorm(d)
        self.head = torch.nn.Linear(d, classes)

    def forward(self, d):
        x = self.patch_embed(_unwrap(d)).flatten(2).transpose(1, 2)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1) + self.pos_embed
        for blk in self.blocks:
            x = blk(x)
        return self.head(self.norm(x)[:, 0])


def init_randn(model: torch.nn.Module, seed: int) -> None:
    """weights ~ N(0, 1/fan_in), biases and embeddings small, LayerNorm left at identity"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "norm" in name:
                continue
            if p.ndim >= 2 and "embed" not in name and "token" not in name:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) / fan_in ** 0.5)
            else:
                p.copy_(0.02 * torch.randn(p.shape, generator=g))


class _BasicBlock(torch.nn.Module):
    def __init__(self, c_in, c_out, stride):
        super().__init__()
        self.conv1 = torch.nn.Conv2d(c_in, c_out, 3, stride=stride, padding=1, bias=False)
        self.bn1 = torch.nn.BatchNorm2d(c_out)
        self.conv2 = torch.nn.Conv2d(c_out, c_out, 3, padding=1, bias=False)
        self.bn2 = torch.nn.BatchNorm2d(c_out)
        self.downsample = None
        if stride != 1 or c_in != c_out:
            self.downsample = torch.nn.Sequential(torch.nn.Conv2d(c_in, c_out, 1, stride=stride, bias=False),
                                                  torch.nn.BatchNorm2d(c_out))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = torch.relu(self.bn1(self.conv1(x)))
        return torch.relu(self.bn2(self.conv2(y)) + idt)


class ResNet18(torch.nn.Module):
    """torchvision resnet18 layout and layer names; decomposable layers (BASELINE configs[0]):
    layer{2,3,4}.0.downsample.0 (1x1, stride 2: SURVEY quirk 5) and fc."""

    def __init__(self, classes=1000, width=64):
        super().__init__()
        w = width
        self.conv1 = torch.nn.Conv2d(3, w, 7, stride=2, padding=3, bias=False)
        self.bn1 = torch.nn.BatchNorm2d(w)
        self.maxpool = torch.nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = torch.nn.Sequential(_BasicBlock(w, w, 1), _BasicBlock(w, w, 1))
        self.layer2 = torch.nn.Sequential(_BasicBlock(w, 2 * w, 2), _BasicBlock(2 * w, 2 * w, 1))
        self.layer3 = torch.nn.Sequential(_BasicBlock(2 * w, 4 * w, 2), _BasicBlock(4 * w, 4 * w, 1))
        self.layer4 = torch.nn.Sequential(_BasicBlock(4 * w, 8 * w, 2), _BasicBlock(8 * w, 8 * w, 1))
        self.fc = torch.nn.Linear(8 * w, classes)

    def forward(self, d):
        x = self.maxpool(torch.relu(self.bn1(self.conv1(_unwrap(d)))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(x.mean(dim=(2, 3)))
