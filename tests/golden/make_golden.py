"""Regenerate the golden fixtures under tests/golden/ from the CPU oracle.

The reference package cannot be imported in this image (Python >= 3.12 syntax, missing
torchvision / lightning / jaxtyping / beartype), and its tests hold no golden vectors, so the expected outputs
come from `oracle/`, whose transform functions make the same torch calls as the reference text
(src/imagescry/image/transforms.py:58-126, src/imagescry/models/embedding.py:74,149-165).

    python tests/golden/make_golden.py
"""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(Path(__file__).resolve().parent))

import cases  # noqa: E402

from oracle import encoder_oracle, search_oracle, transforms_oracle  # noqa: E402

OUT = Path(__file__).resolve().parent


def main() -> None:
    torch.set_num_threads(1)  # summation order of the float32 reductions is then fixed

    # ---- preprocess: no-resize branch (4x3x30x45) and resize branch (2x3x80x100 -> long side 64)
    small = cases.images_u8((4, 3, 30, 45))
    big = cases.images_u8((2, 3, 80, 100), seed=cases.SEED + 1)
    ref_img = cases.reference_test_image()
    np.savez_compressed(
        OUT / "preprocess.npz",
        small_out=encoder_oracle.preprocess(small, max_side_length=640).numpy(),
        small_mean=small.float().mean(dim=(0, 2, 3)).numpy(),
        small_std=small.float().std(dim=(0, 2, 3)).numpy(),
        big_resized=transforms_oracle.resize(big, 64, side_ref="long").numpy(),
        big_out=encoder_oracle.preprocess(big, max_side_length=64).numpy(),
        ref_resize_16_long=transforms_oracle.resize(ref_img, 16, side_ref="long").numpy(),
        ref_resize_5x7=transforms_oracle.resize(ref_img, (5, 7)).numpy(),
        ref_norm=transforms_oracle.normalize_per_channel(ref_img.float().unsqueeze(0)).numpy(),
    )

    # ---- F.normalize over channels, the reference's embedding_batch fixture shape (3 x 128 x 7 x 10)
    x = torch.randn(3, 128, 7, 10, generator=cases.gen())
    np.savez_compressed(OUT / "l2norm.npz", out=encoder_oracle.l2_normalize_channels(x).numpy())

    # ---- search
    out: dict[str, np.ndarray] = {}
    for name, (n, d, q, k, dtype) in cases.SEARCH_CASES.items():
        bank, queries = cases.search_case(n, d, q, dtype)
        s, i = search_oracle.cosine_topk(bank, queries, k)
        out[f"{name}_scores"], out[f"{name}_indices"] = s, i
        full = np.sort(search_oracle.exact_scores(bank, queries).astype(np.float64), axis=1)[:, ::-1]
        out[f"{name}_min_gap"] = np.array((full[:, :k] - full[:, 1 : k + 1]).min())  # near-tie report
    for dtype, tag in ((torch.float16, "f16"), (torch.float32, "f32")):
        bank, queries = cases.tie_case(dtype)
        s, i = search_oracle.cosine_topk(bank, queries, 50)
        out[f"tie_{tag}_scores"], out[f"tie_{tag}_indices"] = s, i
    np.savez_compressed(OUT / "search.npz", **out)
    for key in sorted(out):
        if key.endswith("min_gap"):
            print(f"{key}: {float(out[key]):.3e}")


if __name__ == "__main__":
    main()
