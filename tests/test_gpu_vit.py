"""Transformer-encoder blocks (isc_gemm_f16, isc_layernorm, isc_attention_f16, isc_patchify_f16, isc_vit_assemble) and
the ViT-B/16 embedder (BASELINE.json configs[4]) on the GPU against float32 torch restatements (oracle/vit_oracle.py).

Tolerances.  Kernel tests feed the float32 restatement the SAME fp16-rounded operands the kernel sees, so what is left
is summation order and the fp16 rounding of an fp16 output: rtol 2e-3 (fp16 has 11 significant bits) with a small
absolute floor.  The model test compares L2-normalised embeddings with the plain float32 oracle at the tolerance the
brief states for the fp16 configuration, 1e-2, and with the operand-rounded oracle at 2e-3."""

from __future__ import annotations

import pytest
import torch
import torch.nn.functional as F

from oracle import transforms_oracle
from oracle.vit_oracle import vit_forward

pytestmark = pytest.mark.gpu


def _gen(seed: int) -> torch.Generator:
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("layout", ["rowmajor", "packed", "packed-tile128", "packed-tile256", "rowmajor-tile256"])
@pytest.mark.parametrize(
    "m,k,n,act,res,out_f32",
    [
        (300, 768, 768, "none", True, True),  # attention projection / fc2 shape: residual, float32 stream
        (197, 768, 2304, "none", False, False),  # fused qkv
        (130, 768, 3072, "gelu", False, False),  # fc1
        (257, 3072, 768, "none", True, True),  # fc2, long reduction
        (1, 64, 4, "none", False, True),  # smallest legal problem: one row, four features
        (129, 128, 132, "gelu", True, False),  # ragged in both tile directions
        (2100, 768, 768, "none", True, True),  # several tiles, ragged last one
        (2304, 3072, 772, "none", True, True),  # 48 K steps, ragged feature tile
        (2049, 192, 2304, "none", False, False),  # 3 K steps (the minimum of the 256-tile kernel), fp16 output
        (5000, 64, 512, "gelu", False, False),  # ONE K step per tile on the streaming kernel, several tiles per chunk
        (70000, 128, 256, "none", True, True),  # 274 token tiles x one feature block: chunks of two tiles
        # fp16 output over chunks of THREE tiles (the streaming kernel's spread epilogue: row blocks written at the tile
        # boundary, held in registers across K steps of the next tile, and written in place during its first K step)
        (150000, 192, 256, "none", False, False),  # 586 token tiles, 3 K steps per tile
        (90000, 768, 512, "gelu", False, False),  # 352 token tiles x two feature blocks, 12 K steps, GELU
    ],
)
def test_gemm_f16(device, m, k, n, act, res, out_f32, layout):
    from imagescry_amd import _lib
    from imagescry_amd.vit import pack_rows, packed_elems, unpack_rows

    pk = layout.startswith("packed")
    tile256 = layout.endswith("tile256")
    if pk and n % 64 and not out_f32:
        pytest.skip("a packed fp16 output needs N % 64 == 0")
    if tile256 and (act != "none" or k < 192):
        pytest.skip("the 256-tile kernel has no activation epilogue and needs three K steps")

    g = _gen(m + k + n)
    a = (torch.randn(m, k, generator=g)).half()
    w = (torch.randn(n, k, generator=g) * 0.05).half()
    bias = torch.randn(n, generator=g)
    r = torch.randn(m, n, generator=g) if res else None
    want = a.double() @ w.double().T + bias.double()
    if act == "gelu":
        want = F.gelu(want)
    if res:
        want = want + r.double()
    ad, wd, bd = (pack_rows(a) if pk else a).to(device), (pack_rows(w) if pk else w).to(device), bias.to(device)
    rd = r.to(device) if res else None
    flags = 0
    if pk:
        flags = _lib.ISC_GEMM_A_PACKED | _lib.ISC_GEMM_W_PACKED | (0 if out_f32 else _lib.ISC_GEMM_OUT_PACKED)
    if tile256:
        flags |= _lib.ISC_GEMM_TILE_256
    if layout.endswith("tile128"):  # "packed" alone takes the streaming kernel wherever N % 256 == 0
        flags |= _lib.ISC_GEMM_TILE_128
    n_out = packed_elems(m, n) if pk and not out_f32 else m * n
    out = torch.full((n_out,), float("nan"), dtype=torch.float32 if out_f32 else torch.float16, device=device)
    st = _lib.load().isc_gemm_f16(ad.data_ptr(), m, k, wd.data_ptr(), n, bd.data_ptr(), _lib.ptr(rd),
                                  _lib.ISC_ACT_GELU if act == "gelu" else _lib.ISC_ACT_NONE, out.data_ptr(),
                                  _lib.ISC_F32 if out_f32 else _lib.ISC_F16, flags, _lib.stream_handle(device))
    _lib.check(st, "isc_gemm_f16")
    out = unpack_rows(out.cpu(), m, n) if pk and not out_f32 else out.cpu().view(m, n)
    got = out.double()
    if out_f32:
        torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4)
    else:
        torch.testing.assert_close(got, want, rtol=2e-3, atol=1e-3)


def test_gemm_f16_rejects_bad_arguments(device):
    from imagescry_amd import _lib

    lib = _lib.load()
    a = torch.zeros(8, 96, dtype=torch.float16, device=device)
    w = torch.zeros(8, 96, dtype=torch.float16, device=device)
    out = torch.zeros(8, 8, dtype=torch.float32, device=device)
    s = _lib.stream_handle(device)
    assert lib.isc_gemm_f16(a.data_ptr(), 8, 96, w.data_ptr(), 8, None, None, 0, out.data_ptr(), _lib.ISC_F32, 0, s) == \
        _lib.ISC_ERR_UNSUPPORTED  # K % 64
    assert lib.isc_gemm_f16(a.data_ptr(), 8, 64, w.data_ptr(), 6, None, None, 0, out.data_ptr(), _lib.ISC_F32, 0, s) == \
        _lib.ISC_ERR_UNSUPPORTED  # N % 4
    assert lib.isc_gemm_f16(a.data_ptr(), 8, 64, w.data_ptr(), 8, None, None, _lib.ISC_ACT_RELU, out.data_ptr(),
                            _lib.ISC_F32, 0, s) == _lib.ISC_ERR_INVALID_ARG
    assert lib.isc_gemm_f16(None, 8, 64, w.data_ptr(), 8, None, None, 0, out.data_ptr(), _lib.ISC_F32, 0, s) == \
        _lib.ISC_ERR_INVALID_ARG
    assert lib.isc_gemm_f16(a.data_ptr(), 8, 64, w.data_ptr(), 8, None, None, 0, out.data_ptr(), _lib.ISC_F32,
                            _lib.ISC_GEMM_OUT_PACKED, s) == _lib.ISC_ERR_UNSUPPORTED  # packed output is fp16 only
    assert lib.isc_gemm_f16(a.data_ptr(), 8, 64, w.data_ptr(), 8, None, None, 0, out.data_ptr(), _lib.ISC_F32, 32, s) == \
        _lib.ISC_ERR_INVALID_ARG  # unknown flag
    assert lib.isc_gemm_f16(a.data_ptr(), 8, 64, w.data_ptr(), 8, None, None, 0, out.data_ptr(), _lib.ISC_F32,
                            _lib.ISC_GEMM_TILE_256, s) == _lib.ISC_ERR_UNSUPPORTED  # one K step only


@pytest.mark.parametrize("rows,d,out_f32,pk", [(37, 768, False, False), (5, 768, True, False), (9, 2048, False, False),
                                                (3, 4, True, False), (300, 768, False, True)])
def test_layernorm(device, rows, d, out_f32, pk):
    from imagescry_amd import _lib
    from imagescry_amd.vit import packed_elems, unpack_rows

    g = _gen(rows * d)
    x = torch.randn(rows, d, generator=g) * 3 + 1.5
    gamma, beta = torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g)
    want = F.layer_norm(x, (d,), gamma, beta, 1e-6)
    xd, gd, bd = x.to(device), gamma.to(device), beta.to(device)
    out = torch.empty(packed_elems(rows, d) if pk else rows * d, dtype=torch.float32 if out_f32 else torch.float16,
                      device=device)
    st = _lib.load().isc_layernorm(xd.data_ptr(), rows, d, d, gd.data_ptr(), bd.data_ptr(), 1e-6, out.data_ptr(),
                                   _lib.ISC_F32 if out_f32 else _lib.ISC_F16, d, int(pk), _lib.stream_handle(device))
    _lib.check(st, "isc_layernorm")
    out = unpack_rows(out, rows, d) if pk else out.view(rows, d)
    if out_f32:
        torch.testing.assert_close(out.cpu(), want, rtol=1e-5, atol=1e-5)
    else:
        torch.testing.assert_close(out.float().cpu(), want, rtol=1e-3, atol=1e-3)


def test_layernorm_strided_rows(device):
    """The final LayerNorm reads only the class-token row of every image: row stride T * D."""
    from imagescry_amd import _lib

    g = _gen(7)
    x = torch.randn(6, 5, 768, generator=g)
    gamma, beta = torch.rand(768, generator=g) + 0.5, torch.randn(768, generator=g)
    want = F.layer_norm(x[:, 0], (768,), gamma, beta, 1e-6)
    xd, gd, bd = x.to(device), gamma.to(device), beta.to(device)
    out = torch.empty((6, 768), dtype=torch.float32, device=device)
    st = _lib.load().isc_layernorm(xd.data_ptr(), 6, 768, 5 * 768, gd.data_ptr(), bd.data_ptr(), 1e-6, out.data_ptr(),
                                   _lib.ISC_F32, 768, 0, _lib.stream_handle(device))
    _lib.check(st, "isc_layernorm")
    torch.testing.assert_close(out.cpu(), want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("pk", [False, True])
@pytest.mark.parametrize("b,t,heads", [(2, 197, 12), (3, 17, 2), (1, 224, 1), (2, 1, 3), (1, 16, 4), (1, 50, 12)])
def test_attention(device, b, t, heads, pk):
    from imagescry_amd import _lib
    from imagescry_amd.vit import pack_rows, packed_elems, unpack_rows

    d = heads * 64
    g = _gen(b * t + heads)
    qkv = (torch.randn(b, t, 3 * d, generator=g) * 1.5).half()
    q, k, v = (z.reshape(b, t, heads, 64).transpose(1, 2).double() for z in qkv.split(d, dim=-1))
    att = torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1)
    want = (att @ v).transpose(1, 2).reshape(b, t, d)
    qd = (pack_rows(qkv.reshape(b * t, 3 * d)) if pk else qkv).to(device)
    out = torch.full((packed_elems(b * t, d) if pk else b * t * d,), float("nan"), dtype=torch.float16, device=device)
    st = _lib.load().isc_attention_f16(qd.data_ptr(), b, t, heads, 64, out.data_ptr(), int(pk), _lib.stream_handle(device))
    _lib.check(st, "isc_attention_f16")
    out = (unpack_rows(out, b * t, d) if pk else out).view(b, t, d)
    # probabilities are rounded to fp16 before the second product and the output is fp16
    torch.testing.assert_close(out.double().cpu(), want, rtol=4e-3, atol=2e-3)


@pytest.mark.parametrize("pk", [False, True])
@pytest.mark.parametrize("b,t,heads", [(30, 197, 12), (43, 17, 7), (23, 209, 12), (600, 5, 1), (260, 33, 1)])
def test_attention_more_heads_than_cus(device, b, t, heads, pk):
    """b * heads > 256: the persistent form (k_attention_f16_p: one 16-wave workgroup per CU walks the (image, head) pairs, the
    next pair's keys / values prefetched under the current round).  Against the float64 softmax AND, bit for bit, against the
    one-workgroup-per-head kernel run on slices of at most 256 pairs."""
    from imagescry_amd import _lib
    from imagescry_amd.vit import pack_rows, packed_elems, unpack_rows

    d = heads * 64
    g = _gen(b * t + heads + 5)
    qkv = (torch.randn(b, t, 3 * d, generator=g) * 1.5).half()
    q, k, v = (z.reshape(b, t, heads, 64).transpose(1, 2).double() for z in qkv.split(d, dim=-1))
    want = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1) @ v).transpose(1, 2).reshape(b, t, d)
    lib = _lib.load()

    def run(x: torch.Tensor) -> torch.Tensor:
        n = x.shape[0]
        xd = (pack_rows(x.reshape(n * t, 3 * d)) if pk else x).to(device)
        out = torch.full((packed_elems(n * t, d) if pk else n * t * d,), float("nan"), dtype=torch.float16, device=device)
        _lib.check(lib.isc_attention_f16(xd.data_ptr(), n, t, heads, 64, out.data_ptr(), int(pk), _lib.stream_handle(device)), "att")
        return (unpack_rows(out, n * t, d) if pk else out).view(n, t, d).cpu()

    got = run(qkv)
    torch.testing.assert_close(got.double(), want, rtol=4e-3, atol=2e-3)
    step = max(1, 256 // heads)  # at most 256 pairs per call: the one-shot kernel
    sliced = torch.cat([run(qkv[i : i + step]) for i in range(0, b, step)])
    assert torch.equal(got, sliced)


def test_attention_limits(device):
    from imagescry_amd import _lib

    lib = _lib.load()
    x = torch.zeros(1, 225, 192, dtype=torch.float16, device=device)
    o = torch.zeros(1, 225, 64, dtype=torch.float16, device=device)
    s = _lib.stream_handle(device)
    assert lib.isc_attention_f16(x.data_ptr(), 1, 225, 1, 64, o.data_ptr(), 0, s) == _lib.ISC_ERR_UNSUPPORTED
    assert lib.isc_attention_f16(x.data_ptr(), 1, 100, 1, 32, o.data_ptr(), 0, s) == _lib.ISC_ERR_UNSUPPORTED


def test_patchify_and_assemble(device):
    from imagescry_amd import _lib

    lib = _lib.load()
    s = _lib.stream_handle(device)
    g = _gen(11)
    b, c, hh, ww, p = 3, 3, 48, 64, 16
    x = torch.randn(b, c, hh, ww, generator=g)
    want = F.unfold(x, kernel_size=p, stride=p).transpose(1, 2).reshape(b * (hh // p) * (ww // p), c * p * p).half()
    xd = x.to(device)
    out = torch.empty(want.shape, dtype=torch.float16, device=device)
    _lib.check(lib.isc_patchify_f16(xd.data_ptr(), b, c, hh, ww, p, out.data_ptr(), 0, s), "isc_patchify_f16")
    assert torch.equal(out.cpu(), want)
    from imagescry_amd.vit import packed_elems, unpack_rows

    outp = torch.zeros(packed_elems(*want.shape), dtype=torch.float16, device=device)
    _lib.check(lib.isc_patchify_f16(xd.data_ptr(), b, c, hh, ww, p, outp.data_ptr(), 1, s), "isc_patchify_f16")
    assert torch.equal(unpack_rows(outp, *want.shape).cpu(), want)

    t, d = 13, 768
    pe = torch.randn(b * (t - 1), d, generator=g)
    cls, pos = torch.randn(d, generator=g), torch.randn(t, d, generator=g)
    want_tok = torch.cat([cls.expand(b, 1, d), pe.reshape(b, t - 1, d)], dim=1) + pos
    tok = torch.empty((b, t, d), dtype=torch.float32, device=device)
    ped, clsd, posd = pe.to(device), cls.to(device), pos.to(device)
    _lib.check(lib.isc_vit_assemble(ped.data_ptr(), clsd.data_ptr(), posd.data_ptr(), b, t, d, tok.data_ptr(), s),
               "isc_vit_assemble")
    assert torch.equal(tok.cpu(), want_tok)


@pytest.mark.parametrize("depth,batch", [(2, 3), (12, 2)])
def test_vit_embedder_matches_oracle(device, depth, batch):
    from imagescry_amd import ImageBatch, ViTB16Embedder, vit

    cfg = vit.ViTConfig(depth=depth)
    sd = vit.make_state_dict(cfg, seed=depth, randomize_affine=True)
    model = ViTB16Embedder(config=cfg, state_dict=sd).to(device)
    images = torch.randint(0, 256, (batch, 3, 224, 224), dtype=torch.uint8, generator=_gen(depth))
    got = model.predict_step(ImageBatch(indices=torch.arange(batch), images=images).to(device))
    assert got.embeddings.shape == (batch, 768, 1, 1)
    x = transforms_oracle.normalize_per_channel(images, min_value=-3, max_value=3)
    with torch.no_grad():
        want = F.normalize(vit_forward(sd, x, eps=cfg.ln_eps), dim=1)
        want16 = F.normalize(vit_forward(sd, x, eps=cfg.ln_eps, round_operands_fp16=True), dim=1)
    e = got.embeddings.reshape(batch, 768).cpu()
    assert (e - want).abs().max().item() < 1e-2  # the fp16 tolerance of the brief
    assert (e - want16).abs().max().item() < 2e-3  # against the same operand rounding: implementation error only
    assert torch.allclose(e.norm(dim=1), torch.ones(batch), atol=1e-5)


def test_vit_preprocess_resizes_to_the_token_grid(device):
    from imagescry_amd import ViTB16Embedder, vit

    cfg = vit.ViTConfig(depth=1)
    model = ViTB16Embedder(config=cfg).to(device)
    images = torch.randint(0, 256, (2, 3, 100, 160), dtype=torch.uint8, generator=_gen(4))
    x = model.preprocess(images.to(device))
    assert x.shape == (2, 3, 224, 224)
    want = transforms_oracle.normalize_per_channel(transforms_oracle.resize(images, (224, 224)), min_value=-3, max_value=3)
    torch.testing.assert_close(x.cpu(), want, rtol=1e-4, atol=1e-4)
    assert model(x).shape == (2, 768, 1, 1)
    with pytest.raises(ValueError):
        model(torch.zeros(1, 3, 100, 100, device=device))
