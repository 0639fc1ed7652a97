"""Principal component analysis on MI355X: fit and projection.

Same constructor arguments, attributes and behaviour as the reference `PCA`
(src/imagescry/models/decomposition.py:11-180).  `fit` keeps everything sample-sized on the GPU -- float64 feature sums,
centred + transposed chunks and their Gram matrices on the f32 matrix cores -- and leaves the host the `F x F`
symmetric eigenproblem, where the reference runs an SVD of the whole `[N, F]` matrix on the host.  `transform` /
`forward` -- the step the reference applies to every embedding batch right after the hot path
(src/imagescry/models/pipelines.py:76-84) -- runs in the HIP kernel behind `isc_linear_centered`:
`(x - feature_means) @ component_vectors` with the centring done before the product.
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
        # The same three argument checks as the reference constructor (decomposition.py:46-52: ValueError for each), worded
        # here; callers match on the exception type.
        if min_num_components < 1:
            raise ValueError(f"PCA needs min_num_components >= 1 (got {min_num_components})")
        if max_num_components is not None and max_num_components < min_num_components:
            raise ValueError(
                f"max_num_components={max_num_components} is below min_num_components={min_num_components}"
            )
        if not 0.0 <= min_explained_variance <= 1.0:
            raise ValueError(f"min_explained_variance is a fraction in [0, 1] (got {min_explained_variance})")
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
        # contract string: the reference prints the two sizes, or "not fitted" for both (decomposition.py:73-76)
        shown = (self.num_features, self.num_components) if self.fitted else ("not fitted", "not fitted")
        return f"{type(self).__name__}(num_features={shown[0]}, num_components={shown[1]})"

    # ------------------------------------------------------------------ fit (device + an F x F eigenproblem)
    GRAM_CHUNK_ROWS = 32768  # samples per Gram launch: each chunk's sum is float32, the chunks are added in float64

    def fit(self, x: Tensor) -> "PCA":
        """Principal axes of the centred (not scaled) rows and how many of them the constructor arguments ask for.

        `x` is used as float32 (a float64 input is rounded once, and `feature_means` is float32 whatever the input type).
        The reference centres `x` and takes `torch.linalg.svd` of the `[N, F]` matrix on the host
        (decomposition.py:118-146).  Here everything N-sized runs on the GPU, in three kernels: per-feature sums in
        float64 (`isc_feature_sums`), the centred rows written transposed chunk by chunk (`isc_center_transpose`), and
        the `F x F` Gram matrix of every chunk on the f32 matrix cores (`isc_gram_rows`), added up in float64.  The
        eigenvectors of the Gram matrix ARE the right singular vectors, its eigenvalues the squared singular values, so
        the one thing left for the host is the `F x F` symmetric eigenproblem (LAPACK, float64).  Outputs as the
        reference: `feature_means [1, F]`, `explained_variance [min(N, F)]`, `component_vectors [F, K]`, K chosen by
        the same rule (decomposition.py:133-142).  A principal axis is defined up to its sign; the sign is fixed here by
        making each component's largest-magnitude entry positive.
        """
        if not isinstance(x, Tensor) or not x.dtype.is_floating_point or x.ndim != 2:
            raise TypeError("x must be a floating point [num_samples, num_features] tensor")
        num_samples, num_features = x.shape
        if num_samples < 2:
            raise ValueError(f"num_samples must be at least 2, got {num_samples}")
        if x.device.type != "cuda":
            # The reference is always called as `PCA(...).fit(host_samples)` (decomposition.py:94-148): a host tensor goes
            # to the model's device, or -- for a model that has not been placed yet -- to the current HIP device.  That is a
            # copy, not a CPU computation: without a HIP device the call raises below.
            if self._device.type == "cuda":
                x = x.to(self._device)
            elif torch.cuda.is_available():
                x = x.to(torch.device("cuda", torch.cuda.current_device()))
        _lib.require_device(x, "x")  # no CPU fallback
        dev = x.device
        lib = _lib.load()
        xf = x.detach().float()
        if xf.stride(1) != 1:
            xf = xf.contiguous()
        n, f = num_samples, num_features
        fpad = (f + 3) // 4 * 4
        with torch.cuda.device(dev):
            stream = _lib.stream_handle(dev)
            need = _lib.c_size_t()
            _lib.check(lib.isc_feature_sums_workspace_bytes(n, f, need), "isc_feature_sums_workspace_bytes")
            ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
            sums = torch.empty(f, dtype=torch.float64, device=dev)
            _lib.check(lib.isc_feature_sums(xf.data_ptr(), n, f, xf.stride(0), sums.data_ptr(), ws.data_ptr(), ws.numel(),
                                            stream), "isc_feature_sums")
            mean = (sums / n).float()  # what the reference subtracts: a float32 mean
            # rows per chunk: F * rows < 2^31 (32-bit element offsets in the kernel), whole K steps of 32 samples
            chunk = max(32, min(self.GRAM_CHUNK_ROWS, ((2**31 - 1) // fpad) // 32 * 32))
            ld = (min(chunk, n) + 31) // 32 * 32
            xt = torch.empty((fpad, ld), dtype=torch.float32, device=dev)
            gram_chunk = torch.empty((fpad, fpad), dtype=torch.float32, device=dev)
            gram = torch.zeros((fpad, fpad), dtype=torch.float64, device=dev)
            for r0 in range(0, n, chunk):
                rows = min(chunk, n - r0)
                ldn = (rows + 31) // 32 * 32
                blk = xf[r0 : r0 + rows]
                _lib.check(lib.isc_center_transpose(blk.data_ptr(), rows, f, xf.stride(0), mean.data_ptr(), xt.data_ptr(),
                                                    fpad, ldn, stream), "isc_center_transpose")
                _lib.check(lib.isc_gram_rows(xt.data_ptr(), fpad, ldn, gram_chunk.data_ptr(), stream), "isc_gram_rows")
                gram += gram_chunk.double()
        g = gram[:f, :f].cpu()
        g = (g + g.T) * 0.5
        evals, evecs = torch.linalg.eigh(g)  # ascending; float64
        evals = torch.flip(evals, dims=(0,)).clamp_min(0.0)
        evecs = torch.flip(evecs, dims=(1,))
        lead = evecs.abs().argmax(dim=0)
        evecs = evecs * torch.sign(evecs[lead, torch.arange(f)]).masked_fill_(evecs[lead, torch.arange(f)] == 0, 1.0)
        rank = min(n, f)  # the reference's SVD returns min(N, F) singular values
        # From here on in float32, as the reference computes it (s ** 2 / (n - 1), its sum, the ratio and the cumulative sum
        # are all float32 tensors there, decomposition.py:124-128): with a threshold near 1 the count of components depends
        # on the rounding of the tail of that float32 cumsum, so the same arithmetic is used.  Accuracy floor: the Gram
        # route squares the condition number, eigenvalues below ~1e-7 of the largest are rounding noise (clamped at 0).
        eigenvalues = (evals[:rank] / (n - 1)).float()
        explained = eigenvalues / torch.sum(eigenvalues)
        cumulative = torch.cumsum(explained, dim=0)
        needed = int((cumulative < self.min_explained_variance).sum().item()) + 1
        num_components = max(self.min_num_components, needed)
        if self.max_num_components is not None:
            num_components = min(self.max_num_components, num_components)
        num_components = min(num_components, f)
        self._num_features = f
        self._num_components = num_components
        self.feature_means = mean.reshape(1, f)
        self.explained_variance = explained.to(dev)
        self.component_vectors = evecs[:, :num_components].float().contiguous().to(dev)
        self._fitted = True
        self.hparams.update({"num_features": f, "num_components": num_components})
        self._device = dev
        self._packed = None
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
