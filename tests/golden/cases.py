"""Seeded input generators shared by the golden-fixture script and the tests.

Inputs are regenerated from `torch.Generator().manual_seed(seed)` on the CPU (same torch build here and on the
GPU box), so the committed fixtures only carry the expected OUTPUTS.  Seed 1234 mirrors the reference's tests
(tests/test_models/test_embedding.py:13, tests/test_image/conftest.py:17-18,32).
"""

from __future__ import annotations

import torch

SEED = 1234


def gen(seed: int = SEED) -> torch.Generator:
    return torch.Generator().manual_seed(seed)


def images_u8(shape: tuple[int, ...], seed: int = SEED) -> torch.Tensor:
    return torch.randint(0, 256, shape, dtype=torch.uint8, generator=gen(seed))


def reference_test_image() -> torch.Tensor:
    """The reference's `image_tensor` fixture: seed 1234, uint8 3x30x45 (tests/test_image/conftest.py:28-33)."""
    torch.manual_seed(1234)
    return torch.randint(low=0, high=256, size=(3, 30, 45), dtype=torch.uint8)


def search_case(n: int, d: int, q: int, dtype: torch.dtype, seed: int = SEED) -> tuple[torch.Tensor, torch.Tensor]:
    """Random unit-norm bank `[n, d]` and raw queries `[q, d]`, both cast to `dtype`."""
    g = gen(seed)
    bank = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1).to(dtype)
    queries = torch.randn(q, d, generator=g).to(dtype)
    return bank, queries


def tie_case(dtype: torch.dtype, d: int = 128) -> tuple[torch.Tensor, torch.Tensor]:
    """Bank of 24 distinct unit vectors each repeated 40 times (interleaved) -> exact score ties."""
    g = gen(77)
    base = torch.nn.functional.normalize(torch.randn(24, d, generator=g), dim=1).to(dtype)
    bank = base.repeat(40, 1)  # row i is base[i % 24]
    queries = torch.cat([base[:5].float() * 3.0, torch.randn(6, d, generator=g)]).to(dtype)
    return bank, queries


# name -> (n, d, q, k, dtype)
SEARCH_CASES = {
    "f16_n4096_d768_q64": (4096, 768, 64, 10, torch.float16),
    "f32_n4096_d768_q64": (4096, 768, 64, 10, torch.float32),
    "f16_n20000_d768_q37": (20000, 768, 37, 10, torch.float16),
    "f32_n5001_d96_q3": (5001, 96, 3, 7, torch.float32),
    "f16_n300_d64_q300": (300, 64, 300, 25, torch.float16),
}
