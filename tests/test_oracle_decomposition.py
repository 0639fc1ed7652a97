"""The reference's PCA tests (tests/test_models/test_decomposition.py:42-124), re-run against the oracle's
restatement -- seeds, distributions and expected component counts are the reference's.  The product's `fit` and
`transform` run on the GPU only (tests/test_gpu_decomposition.py re-runs the same cases there); what is checked here is
the oracle, the host-side argument checks and that a host tensor is refused (no CPU fallback)."""

from __future__ import annotations

import pytest
import torch
from torch.distributions import MultivariateNormal

from imagescry_amd import PCA
from oracle import decomposition_oracle

NUM_SAMPLES = 1_000
FEATURE_LOCS = torch.tensor([0.0, 1.0, -1.0, 0.0])
NUM_FEATURES = 4


def uncorrelated_features() -> torch.Tensor:
    torch.manual_seed(1234)
    return MultivariateNormal(loc=FEATURE_LOCS, covariance_matrix=torch.eye(NUM_FEATURES)).sample((NUM_SAMPLES,))


def correlated_features() -> torch.Tensor:
    torch.manual_seed(1234)
    cov = torch.tensor([[1.0, 0.5, 0.0, 0.0], [0.5, 1.0, 0.0, 0.0], [0.0, 0.0, 1.0, -0.5], [0.0, 0.0, -0.5, 1.0]])
    return MultivariateNormal(loc=FEATURE_LOCS, covariance_matrix=cov).sample((NUM_SAMPLES,))


CASES = [(uncorrelated_features, mev, k) for mev, k in [(0.2, 1), (0.4, 2), (0.6, 3), (1.0, 4)]] + [
    (correlated_features, mev, k) for mev, k in [(0.2, 1), (0.4, 2), (0.6, 2), (0.8, 3), (1.0, 4)]
]


@pytest.mark.parametrize("make,min_explained_variance,expected", CASES)
def test_oracle_pca_matches_reference_expectations(make, min_explained_variance: float, expected: int) -> None:
    x = make()
    fitted = decomposition_oracle.fit(x, min_explained_variance=min_explained_variance)
    assert fitted.feature_means.size(1) == NUM_FEATURES and len(fitted.explained_variance) == NUM_FEATURES
    projected = fitted.transform(x)
    assert projected.shape == (NUM_SAMPLES, expected)
    assert fitted.explained_variance[:expected].sum().item() >= min_explained_variance - 1e-6
    if expected > 1:
        corr = torch.abs(torch.corrcoef(projected.T))
        assert torch.all(torch.tril(corr, diagonal=-1) <= 1e-4)


def test_constructor_and_transform_errors() -> None:
    with pytest.raises(ValueError):
        PCA(min_num_components=0)
    with pytest.raises(ValueError):
        PCA(min_num_components=3, max_num_components=2)
    with pytest.raises(ValueError):
        PCA(min_explained_variance=1.5)
    with pytest.raises(RuntimeError, match="not fitted"):
        PCA().transform(torch.zeros(4, 4))
    with pytest.raises(ValueError):
        PCA().fit(torch.zeros(1, 4))
    pca = PCA(min_explained_variance=0.5)
    assert not pca.fitted and "not fitted" in repr(pca)
    from imagescry_amd._lib import HipLibraryError

    with pytest.raises(HipLibraryError):  # fit runs on the GPU: a host tensor is refused, never computed on the host
        pca.fit(uncorrelated_features())
    assert not pca.fitted
