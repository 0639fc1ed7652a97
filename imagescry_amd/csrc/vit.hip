// Transformer-encoder blocks in float16 on the matrix cores (include/imagescry_hip.h: isc_gemm_f16, isc_layernorm,
// isc_attention_f16, isc_patchify_f16, isc_vit_assemble).  Used by the ViT-B/16 embedder (BASELINE.json configs[4]).
//
// Numerics: every product is fp16 x fp16 accumulated in float32 (v_mfma_f32_16x16x32_f16); the residual stream,
// LayerNorm statistics, softmax statistics, bias and GELU are float32.  Only GEMM operands are rounded to fp16.
#include "isc_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }

// ---------------------------------------------------------------------------------------------------------
// out[m][n] = act( sum_k a[m][k] * w[n][k] + bias[n] ) + residual[m][n]
//
// The weight tile is the MFMA "A" operand (rows = output features) and the token tile the "B" operand (columns =
// tokens): a lane then owns four CONSECUTIVE output features of one token, i.e. one 8-byte (fp16) or 16-byte (f32)
// store, and bias / residual are 16-byte loads.  One K step is 64 halves (128 bytes) of every row; LDS tiles are
// [rows][128 B] with the 16-byte chunks XOR-swizzled by (row >> 1) & 7, double buffered, one barrier per K step,
// the next step prefetched into registers while the current one is on the matrix cores.
struct GemmParams {
    const _Float16* a;
    const _Float16* w;
    const float* bias;
    const float* res;
    void* out;
    long long M;
    int N, K, ksteps, act, out_f32;
};

template <int TN, int TM>
__global__ __launch_bounds__(256, 2) void k_gemm_f16(const GemmParams p) {
    constexpr int WN = TN / 64;
    constexpr int NA = TN * 8 / 256;
    constexpr int NB = TM * 8 / 256;
    constexpr int A_BYTES = TN * 128;
    constexpr int B_BYTES = TM * 128;
    constexpr int BUF_BYTES = A_BYTES + B_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wn = wave % WN;
    const int wm = wave / WN;
    const int n_tiles = (p.N + TN - 1) / TN;
    const int n0 = (int)(blockIdx.x % n_tiles) * TN;  // feature tile fastest: one token tile is re-read back to back
    const long long m0 = (long long)(blockIdx.x / n_tiles) * TM;

    const int srow = tid >> 3;
    const int lchunk = (tid & 7) ^ ((srow >> 1) & 7);
    const _Float16* a_ptr[NA];
    const _Float16* b_ptr[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) a_ptr[i] = p.w + (size_t)min(n0 + srow + 32 * i, p.N - 1) * p.K + lchunk * 8;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        long long m = m0 + srow + 32 * i;
        if (m > p.M - 1) m = p.M - 1;
        b_ptr[i] = p.a + (size_t)m * p.K + lchunk * 8;
    }

    const int frow = lane & 15;
    const int fg = lane >> 4;
    const int fsw = (lane >> 1) & 7;
    int foff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) foff[kk] = frow * 128 + (((kk * 4 + fg) ^ fsw) << 4);

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int n = n0 + wn * 64 + mi * 16 + fg * 4;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias && n < p.N) v = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = v;
    }

    u32x4 sa[NA], sb[NB];
    auto load_step = [&](int ks) {
#pragma unroll
        for (int i = 0; i < NA; ++i) sa[i] = *reinterpret_cast<const u32x4*>(a_ptr[i] + ks * 64);
#pragma unroll
        for (int i = 0; i < NB; ++i) sb[i] = *reinterpret_cast<const u32x4*>(b_ptr[i] + ks * 64);
    };
    auto store_step = [&](int buf) {
        unsigned char* a = lds + buf * BUF_BYTES + tid * 16;
        unsigned char* b = a + A_BYTES;
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<u32x4*>(a + 4096 * i) = sa[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<u32x4*>(b + 4096 * i) = sb[i];
    };

    load_step(0);
    store_step(0);
    __syncthreads();

    for (int ks = 0; ks < p.ksteps; ++ks) {
        const int buf = ks & 1;
        const bool more = ks + 1 < p.ksteps;
        if (more) load_step(ks + 1);
        const unsigned char* a_img = lds + buf * BUF_BYTES + wn * 64 * 128;
        const unsigned char* b_img = lds + buf * BUF_BYTES + A_BYTES + wm * 64 * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            u32x4 a[4], b[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) a[mi] = *reinterpret_cast<const u32x4*>(a_img + mi * 2048 + foff[kk]);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) b[ni] = *reinterpret_cast<const u32x4*>(b_img + ni * 2048 + foff[kk]);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a[mi]),
                                                                         __builtin_bit_cast(half8, b[ni]), acc[mi][ni],
                                                                         0, 0, 0);
        }
        if (more) store_step(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const long long m = m0 + wm * 64 + ni * 16 + frow;
        if (m >= p.M) continue;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int n = n0 + wn * 64 + mi * 16 + fg * 4;
            if (n >= p.N) continue;  // N % 4 == 0
            f32x4 v = acc[mi][ni];
            if (p.act == ISC_ACT_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
            }
            const size_t o = (size_t)m * p.N + n;
            if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + o);
            if (p.out_f32) {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + o) = v;
            } else {
                half4 h = half4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                *reinterpret_cast<half4*>(reinterpret_cast<_Float16*>(p.out) + o) = h;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm over the last axis, one wave per row: y = (x - mean) / sqrt(var + eps) * gamma + beta, biased variance,
// float32 statistics (two passes over registers).  D % 4 == 0, D <= 2048.
template <bool OUT_F32, int NV>
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, long long rows, int D, long long ldx,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   float eps, void* __restrict__ y, long long ldy) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + (size_t)row * ldx;
    const int nvec = D >> 2;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < nvec) v[i] = *reinterpret_cast<const f32x4*>(xr + 4 * c);
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = isc_wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = v[i][r] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = 1.f / sqrtf(isc_wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c >= nvec) continue;
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + 4 * c);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (v[i][r] - mean) * rstd * g[r] + b[r];
        if (OUT_F32) {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(y) + (size_t)row * ldy + 4 * c) = o;
        } else {
            half4 h = half4{(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
            *reinterpret_cast<half4*>(reinterpret_cast<_Float16*>(y) + (size_t)row * ldy + 4 * c) = h;
        }
    }
}

template <bool OUT_F32>
void launch_layernorm(int nv, dim3 grid, hipStream_t s, const float* x, long long rows, int D, long long ldx,
                      const float* gamma, const float* beta, float eps, void* y, long long ldy) {
    // NV = 16-byte vectors per lane: the smallest instantiation that covers D keeps the register count (and so the
    // number of rows in flight per CU) where this bandwidth-bound kernel needs it
    if (nv <= 1) hipLaunchKernelGGL((k_layernorm<OUT_F32, 1>), grid, dim3(256), 0, s, x, rows, D, ldx, gamma, beta, eps, y, ldy);
    else if (nv <= 2) hipLaunchKernelGGL((k_layernorm<OUT_F32, 2>), grid, dim3(256), 0, s, x, rows, D, ldx, gamma, beta, eps, y, ldy);
    else if (nv <= 3) hipLaunchKernelGGL((k_layernorm<OUT_F32, 3>), grid, dim3(256), 0, s, x, rows, D, ldx, gamma, beta, eps, y, ldy);
    else if (nv <= 4) hipLaunchKernelGGL((k_layernorm<OUT_F32, 4>), grid, dim3(256), 0, s, x, rows, D, ldx, gamma, beta, eps, y, ldy);
    else hipLaunchKernelGGL((k_layernorm<OUT_F32, 8>), grid, dim3(256), 0, s, x, rows, D, ldx, gamma, beta, eps, y, ldy);
}

// ---------------------------------------------------------------------------------------------------------
// Multi-head self-attention for short sequences (T <= 224, head size 64), one workgroup per (image, head).
//
// qkv [B, T, 3 * D] fp16 with D = heads * 64: query / key / value of head h at columns h*64, D + h*64, 2D + h*64.
// out [B, T, D] fp16.
//
// Keys (row-major) and values (TRANSPOSED) of the head sit in LDS.  Each wave takes 16 queries at a time:
//   S^T[key][query] = K . Q^T / 8      14 key blocks x 2 MFMAs; a lane then holds ONE query (lane & 15) and 56 keys
//   softmax over keys                   registers + two cross-lane steps; statistics in float32
//   O^T[d][query]  = V^T . P^T         the contraction runs over keys, and the order in which a lane's 8 k-slots map
//                                       to keys is free as long as both operands agree: slot (g, j) <-> key
//                                       32 ks + 4 g + j (j < 4) or 32 ks + 16 + 4 g + (j - 4).  With that mapping the
//                                       P^T operand is exactly what the lane already holds after the softmax, so the
//                                       probabilities never leave registers; V^T is read as two 8-byte LDS loads.
constexpr int ATT_TMAX = 224;
constexpr int ATT_KSTRIDE = 72;   // halves per key row (64 + 8 pad)
constexpr int ATT_VSTRIDE = 232;  // halves per value^T row (224 + 8 pad)
constexpr int ATT_THREADS = 512;

__global__ __launch_bounds__(ATT_THREADS) void k_attention_f16(const _Float16* __restrict__ qkv, int T, int heads,
                                                                _Float16* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) _Float16 Ks[ATT_TMAX * ATT_KSTRIDE];
    __shared__ __attribute__((aligned(16))) _Float16 Vt[64 * ATT_VSTRIDE];
    const int b = blockIdx.x / heads;
    const int h = blockIdx.x - b * heads;
    const int D = heads * 64;
    const size_t row_stride = (size_t)3 * D;
    const _Float16* base = qkv + (size_t)b * T * row_stride + h * 64;
    const int tid = threadIdx.x;

    for (int i = tid; i < ATT_TMAX * 8; i += ATT_THREADS) {
        const int t = i >> 3, c = i & 7;
        half8 kv = half8{0, 0, 0, 0, 0, 0, 0, 0};
        half8 vv = kv;
        if (t < T) {
            kv = *reinterpret_cast<const half8*>(base + (size_t)t * row_stride + D + c * 8);
            vv = *reinterpret_cast<const half8*>(base + (size_t)t * row_stride + 2 * D + c * 8);
        }
        *reinterpret_cast<half8*>(&Ks[t * ATT_KSTRIDE + c * 8]) = kv;
#pragma unroll
        for (int j = 0; j < 8; ++j) Vt[(c * 8 + j) * ATT_VSTRIDE + t] = vv[j];
    }
    __syncthreads();

    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int qi = lane & 15;
    const int g = lane >> 4;
    const int nqb = (T + 15) >> 4;
    for (int qb = wave; qb < nqb; qb += ATT_THREADS / 64) {
        const int tq = qb * 16 + qi;
        const _Float16* qrow = base + (size_t)min(tq, T - 1) * row_stride;
        half8 qf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            qf[kk] = *reinterpret_cast<const half8*>(qrow + (kk * 4 + g) * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[kk][j] = qf[kk][j] * (_Float16)0.125f;  // 1/sqrt(64): exact scaling
        }
        f32x4 s[14];
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 14; ++kb) {
            f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const half8 kf = *reinterpret_cast<const half8*>(&Ks[(kb * 16 + qi) * ATT_KSTRIDE + (kk * 4 + g) * 8]);
                a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[kk], a, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (kb * 16 + g * 4 + r >= T) a[r] = -INFINITY;
                mx = fmaxf(mx, a[r]);
            }
            s[kb] = a;
            __builtin_amdgcn_sched_barrier(0);  // keep the fragment loads of later key blocks from piling up in VGPRs
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 14; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(s[kb][r] - mx);
                s[kb][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;

        f32x4 o[4];
#pragma unroll
        for (int db = 0; db < 4; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 7; ++ks) {
            half8 pf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[r] = (_Float16)s[2 * ks][r];
                pf[4 + r] = (_Float16)s[2 * ks + 1][r];
            }
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const _Float16* vrow = &Vt[(db * 16 + qi) * ATT_VSTRIDE + ks * 32 + g * 4];
                const half4 lo = *reinterpret_cast<const half4*>(vrow);
                const half4 hi = *reinterpret_cast<const half4*>(vrow + 16);
                const half8 vf = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                o[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o[db], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (tq < T) {
            _Float16* orow = out + ((size_t)b * T + tq) * D + h * 64;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                half4 hv;
#pragma unroll
                for (int r = 0; r < 4; ++r) hv[r] = (_Float16)(o[db][r] * inv);
                *reinterpret_cast<half4*>(orow + db * 16 + g * 4) = hv;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// patches[b * gh * gw + (ph * gw + pw)][c * P * P + r * P + s] = fp16(x[b][c][ph * P + r][pw * P + s]); 8 values a thread
__global__ __launch_bounds__(256) void k_patchify_f16(const float* __restrict__ x, int C, int H, int W, int P,
                                                      size_t total8, _Float16* __restrict__ y) {
    const int gw = W / P, gh = H / P;
    const int pv = P / 8;
    const int kdim = C * P * P;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (size_t)gridDim.x * 256) {
        // i enumerates (b, c, row, 8-pixel group) in input order, so loads are coalesced
        const int wg = (int)(i % (W / 8));
        size_t rest = i / (W / 8);
        const int hrow = (int)(rest % H);
        rest /= H;
        const int c = (int)(rest % C);
        const size_t b = rest / C;
        const int ph = hrow / P, r = hrow - ph * P;
        const int pw = wg / pv, s = (wg - pw * pv) * 8;
        const float* src = x + ((b * C + c) * H + hrow) * (size_t)W + wg * 8;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + 4);
        half8 o = half8{(_Float16)v0[0], (_Float16)v0[1], (_Float16)v0[2], (_Float16)v0[3],
                        (_Float16)v1[0], (_Float16)v1[1], (_Float16)v1[2], (_Float16)v1[3]};
        const size_t m = (b * gh + ph) * gw + pw;
        *reinterpret_cast<half8*>(y + m * kdim + (c * P + r) * P + s) = o;
    }
}

// tokens[b][0] = cls + pos[0];  tokens[b][t] = patch_embed[b * (T - 1) + t - 1] + pos[t]
__global__ __launch_bounds__(256) void k_vit_assemble(const float* __restrict__ pe, const float* __restrict__ cls,
                                                      const float* __restrict__ pos, int T, int D, size_t total4,
                                                      float* __restrict__ y) {
    const int dv = D / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % dv);
        const size_t bt = i / dv;
        const int t = (int)(bt % T);
        const size_t b = bt / T;
        f32x4 v = t == 0 ? *reinterpret_cast<const f32x4*>(cls + 4 * c)
                         : *reinterpret_cast<const f32x4*>(pe + (b * (T - 1) + t - 1) * (size_t)D + 4 * c);
        v += *reinterpret_cast<const f32x4*>(pos + (size_t)t * D + 4 * c);
        *reinterpret_cast<f32x4*>(y + i * 4) = v;
    }
}

int grid_for(size_t work_items) {
    const size_t blocks = isc_ceil_div<size_t>(work_items, 256);
    return (int)(blocks < 16384 ? (blocks ? blocks : 1) : 16384);
}

}  // namespace

extern "C" int isc_gemm_f16(const void* a, int64_t M, int K, const void* w, int N, const float* bias,
                            const float* residual, int act, void* out, int out_dtype, void* stream) {
    ISC_REQUIRE(a && w && out && M > 0 && K > 0 && N > 0);
    ISC_REQUIRE(act == ISC_ACT_NONE || act == ISC_ACT_GELU);
    ISC_REQUIRE(out_dtype == ISC_F16 || out_dtype == ISC_F32);
    if (K % 64 != 0 || N % 4 != 0) return ISC_ERR_UNSUPPORTED;
    if (!isc_aligned(a, 16) || !isc_aligned(w, 16) || !isc_aligned(out, 16) || (bias && !isc_aligned(bias, 16)) ||
        (residual && !isc_aligned(residual, 16)))
        return ISC_ERR_ALIGNMENT;
    GemmParams p;
    p.a = reinterpret_cast<const _Float16*>(a);
    p.w = reinterpret_cast<const _Float16*>(w);
    p.bias = bias;
    p.res = residual;
    p.out = out;
    p.M = M;
    p.N = N;
    p.K = K;
    p.ksteps = K / 64;
    p.act = act;
    p.out_f32 = out_dtype == ISC_F32;
    const long long tiles = isc_ceil_div<long long>(M, 128) * isc_ceil_div<long long>(N, 128);
    if (tiles > 0x7fffffffLL) return ISC_ERR_UNSUPPORTED;
    hipStream_t s = isc_stream(stream);
    isc_timing_begin(ISC_KERNEL_GEMM_F16, s);
    hipLaunchKernelGGL((k_gemm_f16<128, 128>), dim3((unsigned)tiles), dim3(256), 0, s, p);
    isc_timing_end(ISC_KERNEL_GEMM_F16, s);
    return isc_launch_status();
}

extern "C" int isc_layernorm(const float* x, int64_t rows, int D, int64_t ldx, const float* gamma, const float* beta,
                             float eps, void* y, int y_dtype, int64_t ldy, void* stream) {
    ISC_REQUIRE(x && gamma && beta && y && rows > 0 && D > 0 && eps >= 0.f);
    ISC_REQUIRE(y_dtype == ISC_F16 || y_dtype == ISC_F32);
    if (D % 4 != 0 || D > 2048) return ISC_ERR_UNSUPPORTED;
    if (ldx < D || ldy < D || ldx % 4 != 0 || ldy % 4 != 0) return ISC_ERR_ALIGNMENT;
    if (!isc_aligned(x, 16) || !isc_aligned(gamma, 16) || !isc_aligned(beta, 16) || !isc_aligned(y, 16))
        return ISC_ERR_ALIGNMENT;
    const long long blocks = isc_ceil_div<long long>(rows, 4);
    if (blocks > 0x7fffffffLL) return ISC_ERR_UNSUPPORTED;
    hipStream_t s = isc_stream(stream);
    const int nv = (D / 4 + 63) / 64;
    if (y_dtype == ISC_F32)
        launch_layernorm<true>(nv, dim3((unsigned)blocks), s, x, (long long)rows, D, (long long)ldx, gamma, beta, eps, y,
                               (long long)ldy);
    else
        launch_layernorm<false>(nv, dim3((unsigned)blocks), s, x, (long long)rows, D, (long long)ldx, gamma, beta, eps, y,
                                (long long)ldy);
    return isc_launch_status();
}

extern "C" int isc_attention_f16(const void* qkv, int B, int T, int heads, int head_dim, void* out, void* stream) {
    ISC_REQUIRE(qkv && out && B > 0 && T > 0 && heads > 0);
    if (head_dim != 64 || T > ATT_TMAX) return ISC_ERR_UNSUPPORTED;
    if (!isc_aligned(qkv, 16) || !isc_aligned(out, 16)) return ISC_ERR_ALIGNMENT;
    if ((long long)B * heads > 0x7fffffffLL) return ISC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_attention_f16, dim3((unsigned)(B * heads)), dim3(ATT_THREADS), 0, isc_stream(stream),
                       reinterpret_cast<const _Float16*>(qkv), T, heads, reinterpret_cast<_Float16*>(out));
    return isc_launch_status();
}

extern "C" int isc_patchify_f16(const float* x, int B, int C, int H, int W, int patch, void* patches, void* stream) {
    ISC_REQUIRE(x && patches && B > 0 && C > 0 && H > 0 && W > 0 && patch > 0);
    if (patch % 8 != 0 || H % patch != 0 || W % patch != 0) return ISC_ERR_UNSUPPORTED;
    if (!isc_aligned(x, 16) || !isc_aligned(patches, 16)) return ISC_ERR_ALIGNMENT;
    const size_t total8 = (size_t)B * C * H * (W / 8);
    hipLaunchKernelGGL(k_patchify_f16, dim3(grid_for(total8)), dim3(256), 0, isc_stream(stream), x, C, H, W, patch, total8,
                       reinterpret_cast<_Float16*>(patches));
    return isc_launch_status();
}

extern "C" int isc_vit_assemble(const float* patch_embed, const float* cls_token, const float* pos_embed, int B, int T,
                                int D, float* tokens, void* stream) {
    ISC_REQUIRE(patch_embed && cls_token && pos_embed && tokens && B > 0 && T > 1 && D > 0);
    if (D % 4 != 0) return ISC_ERR_UNSUPPORTED;
    if (!isc_aligned(patch_embed, 16) || !isc_aligned(cls_token, 16) || !isc_aligned(pos_embed, 16) ||
        !isc_aligned(tokens, 16))
        return ISC_ERR_ALIGNMENT;
    const size_t total4 = (size_t)B * T * (D / 4);
    hipLaunchKernelGGL(k_vit_assemble, dim3(grid_for(total4)), dim3(256), 0, isc_stream(stream), patch_embed, cls_token,
                       pos_embed, T, D, total4, tokens);
    return isc_launch_status();
}
