"""CPU restatement of the reference PCA (test infrastructure, see oracle/__init__.py).

Follows src/imagescry/models/decomposition.py:94-148 (`fit`) and :78-91 (`forward`) call for call; the Lightning
`nn.Parameter` wrappers are the only thing left out.  Pinned by the reference's own tests, which are re-run against
this restatement in tests/test_oracle_decomposition.py (expected component counts, explained variance,
decorrelation of the projected features: tests/test_models/test_decomposition.py:42-124).
"""

from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import Tensor


@dataclass
class FittedPCA:
    feature_means: Tensor  # [1, F]
    explained_variance: Tensor  # [F]
    component_vectors: Tensor  # [F, K]

    @property
    def num_components(self) -> int:
        return self.component_vectors.shape[1]

    def transform(self, x: Tensor) -> Tensor:
        """reference: decomposition.py:91."""
        return torch.matmul(x - self.feature_means, self.component_vectors)


def fit(
    x: Tensor, *, min_num_components: int = 1, max_num_components: int | None = None, min_explained_variance: float = 0.0
) -> FittedPCA:
    """reference: decomposition.py:113-148."""
    num_samples, _ = x.shape
    feature_means = x.mean(dim=0, keepdim=True)
    x_centered = x - feature_means
    _, s, vt = torch.linalg.svd(x_centered)
    eigenvalues = s**2 / (num_samples - 1)
    explained_variance = eigenvalues / torch.sum(eigenvalues)
    cumulative = torch.cumsum(explained_variance, dim=0)
    needed = int(torch.sum(cumulative < min_explained_variance).item() + 1)
    num_components = max(min_num_components, needed)
    if max_num_components is not None:
        num_components = min(max_num_components, num_components)
    return FittedPCA(feature_means, explained_variance, vt[:num_components, :].T)
