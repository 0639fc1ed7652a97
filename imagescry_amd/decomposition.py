"""Principal component analysis with the projection on MI355X.

Same constructor arguments, attributes and behaviour as the reference `PCA`
(src/imagescry/models/decomposition.py:11-180).  `fit` runs the SVD on the host (one-off, LAPACK through torch,
exactly the reference's calls); `transform` / `forward` -- the step the reference applies to every embedding
batch right after the hot path (src/imagescry/models/pipelines.py:76-84) -- runs in the HIP kernel behind
`isc_linear_centered`: `(x - feature_means) @ component_vectors` with the centring done before the product.
"""

from __future__ import annotations

import torch
from torch import Tensor

from imagescry_amd import _lib

__all__ = ["PCA"]


class PCA:
    """Linear projection to a lower dimensional space using the SVD of the centred data."""

    def __init__(
        self,
        *,
        min_num_components: int = 1,
        max_num_components: int | None = None,
        min_explained_variance: float = 0.0,
        num_features: int = 0,
        num_components: int = 0,
    ) -> None:
        # reference: decomposition.py:46-52
        if min_num_components < 1:
            raise ValueError(f"min_num_components must be at least 1, got {min_num_components}")
        if max_num_components is not None and max_num_components < min_num_components:
            raise ValueError(f"max_num_components must be at least {min_num_components}, got {max_num_components}")
        if min_explained_variance < 0.0 or min_explained_variance > 1.0:
            raise ValueError(f"min_explained_variance must be between 0.0 and 1.0, got {min_explained_variance}")
        self.min_num_components = min_num_components
        self.max_num_components = max_num_components
        self.min_explained_variance = min_explained_variance
        self.hparams: dict[str, object] = {
            "min_num_components": min_num_components,
            "max_num_components": max_num_components,
            "min_explained_variance": min_explained_variance,
        }
        self._fitted = False
        self._num_features = 0
        self._num_components = 0
        self.feature_means = torch.empty((1, num_features))
        self.explained_variance = torch.empty((num_features,))
        self.component_vectors = torch.empty((num_features, num_components))
        self._device = torch.device("cpu")
        self._packed: tuple[Tensor, Tensor] | None = None  # (mean [Fpad], weights [Kpad, Fpad]) on the device

    def __repr__(self) -> str:
        num_features = self.num_features if self.fitted else "not fitted"
        num_components = self.num_components if self.fitted else "not fitted"
        return f"{self.__class__.__name__}(num_features={num_features}, num_components={num_components})"

    # ------------------------------------------------------------------ fit (host)
    def fit(self, x: Tensor) -> "PCA":
        """Centre (not scale) the features, SVD, keep the components the constructor arguments ask for
        (reference: decomposition.py:94-148).  `x` may live on any device; the SVD runs on the host."""
        if not isinstance(x, Tensor) or not x.dtype.is_floating_point or x.ndim != 2:
            raise TypeError("x must be a floating point [num_samples, num_features] tensor")
        num_samples, num_features = x.shape
        if num_samples < 2:
            raise ValueError(f"num_samples must be at least 2, got {num_samples}")
        xh = x.detach().cpu()
        self._num_features = num_features
        self.feature_means = xh.mean(dim=0, keepdim=True)
        x_centered = xh - self.feature_means
        _, s, vt = torch.linalg.svd(x_centered)
        eigenvalues = s**2 / (num_samples - 1)
        total_variance = torch.sum(eigenvalues)
        self.explained_variance = eigenvalues / total_variance
        cumulative = torch.cumsum(self.explained_variance, dim=0)
        needed = int(torch.sum(cumulative < self.min_explained_variance).item() + 1)
        num_components = max(self.min_num_components, needed)
        if self.max_num_components is not None:
            num_components = min(self.max_num_components, num_components)
        self._num_components = num_components
        self.component_vectors = vt[:num_components, :].T.contiguous()
        self._fitted = True
        self.hparams.update({"num_features": num_features, "num_components": num_components})
        self._packed = None
        if x.device.type == "cuda":
            self.to(x.device)
        return self

    # ------------------------------------------------------------------ device placement
    def to(self, device: str | torch.device) -> "PCA":
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.feature_means = self.feature_means.to(device)
        self.explained_variance = self.explained_variance.to(device)
        self.component_vectors = self.component_vectors.to(device)
        self._device = device
        self._packed = None
        return self

    @property
    def device(self) -> torch.device:
        return self._device

    def _kernel_operands(self) -> tuple[Tensor, Tensor]:
        """mean padded to a multiple of 32 features, components as [K padded to 4, F padded] rows (zeros in the
        padding, so padded inputs and outputs contribute nothing)."""
        if self._packed is None:
            f, k = self.num_features, self.num_components
            fpad, kpad = (f + 31) // 32 * 32, (k + 3) // 4 * 4
            mean = torch.zeros(fpad, dtype=torch.float32, device=self._device)
            mean[:f] = self.feature_means.reshape(-1).float()
            w = torch.zeros((kpad, fpad), dtype=torch.float32, device=self._device)
            w[:k, :f] = self.component_vectors.T.float()
            self._packed = (mean, w)
        return self._packed

    # ------------------------------------------------------------------ projection (device)
    def forward(self, x: Tensor) -> Tensor:
        """`(x - feature_means) @ component_vectors` (reference: decomposition.py:78-91)."""
        if not isinstance(x, Tensor) or not x.dtype.is_floating_point:
            raise TypeError("x must be a floating point tensor")
        if x.ndim != 2 or x.shape[1] != self.num_features:
            raise ValueError(f"x must have shape [num_samples, {self.num_features}], got {tuple(x.shape)}")
        _lib.require_device(x, "x")
        if self._device != x.device:
            raise ValueError(f"PCA is on {self._device} but the input is on {x.device}; call .to() first")
        n, f = x.shape
        k = self.num_components
        out_dtype = x.dtype
        mean, w = self._kernel_operands()
        fpad, kpad = w.shape[1], w.shape[0]
        xk = x.float()
        if fpad != f:
            xk = torch.nn.functional.pad(xk, (0, fpad - f))
        xk = xk.contiguous()
        out = torch.empty((n, kpad), dtype=torch.float32, device=x.device)
        if n > 0:
            lib = _lib.load()
            with torch.cuda.device(x.device):
                st = lib.isc_linear_centered(
                    xk.data_ptr(), n, fpad, mean.data_ptr(), w.data_ptr(), kpad, None, out.data_ptr(),
                    _lib.stream_handle(x.device),
                )
            _lib.check(st, "isc_linear_centered")
        return out[:, :k].to(out_dtype)

    def __call__(self, x: Tensor) -> Tensor:
        return self.forward(x)

    def transform(self, x: Tensor) -> Tensor:
        """Project the input data (reference: decomposition.py:150-165)."""
        if not self.fitted:
            raise RuntimeError("PCA model not fitted")
        return self(x)

    @property
    def fitted(self) -> bool:
        return self._fitted

    @property
    def num_features(self) -> int:
        return self._num_features

    @property
    def num_components(self) -> int:
        return self._num_components
