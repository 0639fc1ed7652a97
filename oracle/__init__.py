"""CPU oracle for the embed-and-search hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under `oracle/` is product code.  Only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import it, and only as the checker.
The product path (`imagescry_amd/`) never imports this package and fails loudly
when its HIP library is missing.

What it restates (each function cites the reference file:line it follows):

* `transforms_oracle`  -- `normalize_per_channel`, `resize`, `to_4d`
  (reference src/imagescry/image/transforms.py:16-197)
* `encoder_oracle`     -- the `preprocess -> forward -> F.normalize` order of
  `EmbeddingModule.predict_step` (reference src/imagescry/models/embedding.py:57-76,
  149-165) and a plain `torch.nn.functional` ResNet-50 -> 768-d encoder
* `search_oracle`      -- cosine top-k.  The reference has NO search code
  (SURVEY.md section 0 fact 2); the expression is the one `north_star` states.

Pinning status: the reference package cannot be imported in
this image (Python >= 3.12 syntax, torchvision / lightning / jaxtyping / beartype
absent, no network) and holds no golden vectors.  The transforms restatement calls
the same torch primitives the reference calls, argument for argument, and is pinned
by the reference's own property tests re-run in tests/test_oracle_transforms.py
(mean~0 / std~1 at 1e-4, resize output shapes).  Encoder VALUES and the whole search
step are "parity unpinned" by the reference: they are pinned only by this oracle.
"""
