// Preprocess kernels: batch-statistics normalisation, bilinear resize, L2 normalisation.
// All three are HBM-bound byte/float streams; they are written for coalesced 16-byte lanes,
// not for the matrix cores.  Reference arithmetic: src/imagescry/image/transforms.py:58-126 and
// src/imagescry/models/embedding.py:74.
#include "isc_common.h"

namespace {

constexpr int kStatsThreads = 256;
constexpr int kStatsChunk = 16384;  // elements of one plane handled by one workgroup

struct StatPartial {
    double s;   // sum x      (exact integer value for u8 input)
    double ss;  // sum x * x
};

// One workgroup reduces one chunk of one (b, c) plane.  u8 pixels are summed as integers (v_dot4_u32_u8),
// so the partials -- and therefore mean / std -- do not depend on the launch geometry.
__global__ __launch_bounds__(kStatsThreads) void k_stats_partial_u8(const uint8_t* __restrict__ x, int C, int HW,
                                                                    int chunks_per_plane, StatPartial* __restrict__ part,
                                                                    int items_per_channel, int vec_ok) {
    const int item = blockIdx.x;  // (b, chunk)
    const int c = blockIdx.y;
    const int b = item / chunks_per_plane;
    const int chunk = item - b * chunks_per_plane;
    const int begin = chunk * kStatsChunk;
    const int end = min(HW, begin + kStatsChunk);
    const uint8_t* p = x + ((size_t)b * C + c) * HW;
    unsigned s = 0, ss = 0;  // <= 64 * 65025 per thread: no overflow
    if (vec_ok) {
        for (int i = begin + threadIdx.x * 16; i < end; i += kStatsThreads * 16) {
            const uint4 v = *reinterpret_cast<const uint4*>(p + i);
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s = __builtin_amdgcn_udot4(w[j], 0x01010101u, s, false);
                ss = __builtin_amdgcn_udot4(w[j], w[j], ss, false);
            }
        }
    } else {
        for (int i = begin + threadIdx.x; i < end; i += kStatsThreads) {
            const unsigned v = p[i];
            s += v;
            ss += v * v;
        }
    }
    unsigned long long s64 = isc_wave_sum((unsigned long long)s);
    unsigned long long ss64 = isc_wave_sum((unsigned long long)ss);
    __shared__ unsigned long long red[2][kStatsThreads / ISC_WAVE];
    const int wave = threadIdx.x / ISC_WAVE;
    if ((threadIdx.x & 63) == 0) {
        red[0][wave] = s64;
        red[1][wave] = ss64;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long a = 0, q = 0;
        for (int w = 0; w < kStatsThreads / ISC_WAVE; ++w) {
            a += red[0][w];
            q += red[1][w];
        }
        part[(size_t)c * items_per_channel + item] = StatPartial{(double)a, (double)q};
    }
}

__global__ __launch_bounds__(kStatsThreads) void k_stats_partial_f32(const float* __restrict__ x, int C, int HW,
                                                                     int chunks_per_plane, StatPartial* __restrict__ part,
                                                                     int items_per_channel, int vec_ok) {
    const int item = blockIdx.x;
    const int c = blockIdx.y;
    const int b = item / chunks_per_plane;
    const int chunk = item - b * chunks_per_plane;
    const int begin = chunk * kStatsChunk;
    const int end = min(HW, begin + kStatsChunk);
    const float* p = x + ((size_t)b * C + c) * HW;
    double s = 0.0, ss = 0.0;
    if (vec_ok) {
        for (int i = begin + threadIdx.x * 4; i < end; i += kStatsThreads * 4) {
            const float4 v = *reinterpret_cast<const float4*>(p + i);
            s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
            ss += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        }
    } else {
        for (int i = begin + threadIdx.x; i < end; i += kStatsThreads) {
            const double v = p[i];
            s += v;
            ss += v * v;
        }
    }
    s = isc_wave_sum(s);
    ss = isc_wave_sum(ss);
    __shared__ double red[2][kStatsThreads / ISC_WAVE];
    const int wave = threadIdx.x / ISC_WAVE;
    if ((threadIdx.x & 63) == 0) {
        red[0][wave] = s;
        red[1][wave] = ss;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, q = 0;
        for (int w = 0; w < kStatsThreads / ISC_WAVE; ++w) {
            a += red[0][w];
            q += red[1][w];
        }
        part[(size_t)c * items_per_channel + item] = StatPartial{a, q};
    }
}

// One workgroup per channel: fixed-order sum of the partials, then mean and unbiased std in float64.
__global__ __launch_bounds__(kStatsThreads) void k_stats_final(const StatPartial* __restrict__ part, int items_per_channel,
                                                               double n, float* __restrict__ mean,
                                                               float* __restrict__ stdev) {
    const int c = blockIdx.x;
    double s = 0.0, ss = 0.0;
    for (int i = threadIdx.x; i < items_per_channel; i += kStatsThreads) {
        const StatPartial p = part[(size_t)c * items_per_channel + i];
        s += p.s;
        ss += p.ss;
    }
    s = isc_wave_sum(s);
    ss = isc_wave_sum(ss);
    __shared__ double red[2][kStatsThreads / ISC_WAVE];
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x / ISC_WAVE] = s;
        red[1][threadIdx.x / ISC_WAVE] = ss;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, q = 0;
        for (int w = 0; w < kStatsThreads / ISC_WAVE; ++w) {
            a += red[0][w];
            q += red[1][w];
        }
        const double m = a / n;
        double var = (q - a * a / n) / (n - 1.0);  // n == 1 -> NaN, as torch's unbiased std
        if (var < 0.0) var = 0.0;
        mean[c] = (float)m;
        stdev[c] = (float)sqrt(var);
    }
}

__device__ __forceinline__ float norm_clip(float v, float mu, float denom, float lo, float hi) {
    const float r = __fdiv_rn(v - mu, denom);
    return r != r ? r : fminf(fmaxf(r, lo), hi);
}

// 16 pixels per lane: one 16-byte load, four 16-byte stores.  Requires HW % 16 == 0 so a vector never
// straddles two planes (224*224 = 16 * 3136).
__global__ __launch_bounds__(256) void k_normalize_u8_vec16(const uint8_t* __restrict__ x, size_t nvec, int C, int HW,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ stdev, int per_image, float eps,
                                                            float lo, float hi, float* __restrict__ y) {
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
        const size_t e = v * 16;
        const size_t plane = e / HW;
        const int st = per_image ? (int)plane : (int)(plane % C);
        const float mu = mean[st];
        const float denom = stdev[st] + eps;
        const uint4 in = *reinterpret_cast<const uint4*>(x + e);
        const unsigned w[4] = {in.x, in.y, in.z, in.w};
        float4* out = reinterpret_cast<float4*>(y + e);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 o;
            o.x = norm_clip((float)(w[j] & 0xffu), mu, denom, lo, hi);
            o.y = norm_clip((float)((w[j] >> 8) & 0xffu), mu, denom, lo, hi);
            o.z = norm_clip((float)((w[j] >> 16) & 0xffu), mu, denom, lo, hi);
            o.w = norm_clip((float)(w[j] >> 24), mu, denom, lo, hi);
            out[j] = o;
        }
    }
}

// 4 pixels per lane: one 4-byte load, ONE 16-byte store -- a wave reads 256 contiguous bytes and writes one contiguous KiB per
// instruction.  (k_normalize_u8_vec16's four stores per lane each write 64 pieces of 16 bytes 64 bytes apart, every 128-byte
// line four times over: 116 us against this kernel's ~75 at 512 x 3 x 224 x 224; the 16-pixel form is no longer launched.)
__global__ __launch_bounds__(256) void k_normalize_u8_vec4(const uint8_t* __restrict__ x, size_t nvec, int C, int HW,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ stdev, int per_image, float eps,
                                                           float lo, float hi, float* __restrict__ y) {
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
        const size_t e = v * 4;
        const size_t plane = e / HW;
        const int st = per_image ? (int)plane : (int)(plane % C);
        const float mu = mean[st];
        const float denom = stdev[st] + eps;
        const unsigned w = *reinterpret_cast<const unsigned*>(x + e);
        float4 o;
        o.x = norm_clip((float)(w & 0xffu), mu, denom, lo, hi);
        o.y = norm_clip((float)((w >> 8) & 0xffu), mu, denom, lo, hi);
        o.z = norm_clip((float)((w >> 16) & 0xffu), mu, denom, lo, hi);
        o.w = norm_clip((float)(w >> 24), mu, denom, lo, hi);
        *reinterpret_cast<float4*>(y + e) = o;
    }
}

__global__ __launch_bounds__(256) void k_normalize_f32_vec4(const float* __restrict__ x, size_t nvec, int C, int HW,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ stdev, int per_image, float eps,
                                                            float lo, float hi, float* __restrict__ y) {
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
        const size_t e = v * 4;
        const size_t plane = e / HW;
        const int st = per_image ? (int)plane : (int)(plane % C);
        const float mu = mean[st];
        const float denom = stdev[st] + eps;
        const float4 in = *reinterpret_cast<const float4*>(x + e);
        float4 o;
        o.x = norm_clip(in.x, mu, denom, lo, hi);
        o.y = norm_clip(in.y, mu, denom, lo, hi);
        o.z = norm_clip(in.z, mu, denom, lo, hi);
        o.w = norm_clip(in.w, mu, denom, lo, hi);
        *reinterpret_cast<float4*>(y + e) = o;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_normalize_scalar(const T* __restrict__ x, size_t n, int C, int HW,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ stdev, int per_image, float eps,
                                                          float lo, float hi, float* __restrict__ y) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t plane = e / HW;
        const int st = per_image ? (int)plane : (int)(plane % C);
        y[e] = norm_clip((float)x[e], mean[st], stdev[st] + eps, lo, hi);
    }
}

// The same normalisation written channels-last for the convolution stems: y[b][hw][0..C) = norm_clip(x[b][c][hw]),
// channels C..3 zero.  PX consecutive pixels per thread (PX = 4 needs HW % 4 == 0): one PX-wide load per plane,
// PX 16-byte stores (a wave writes 64 * PX * 16 contiguous bytes).
template <typename T, int PX>
__global__ __launch_bounds__(256) void k_normalize_nhwc4(const T* __restrict__ x, size_t ngroups, int C, int HW,
                                                         const float* __restrict__ mean, const float* __restrict__ stdev,
                                                         int per_image, float eps, float lo, float hi,
                                                         float* __restrict__ y) {
    const size_t groups_per_image = (size_t)HW / PX;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (size_t)gridDim.x * blockDim.x) {
        const size_t b = g / groups_per_image;
        const size_t hw = (g - b * groups_per_image) * PX;
        float o[PX][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < C) {
                const int st = per_image ? (int)(b * C + c) : c;
                const float mu = mean[st];
                const float denom = stdev[st] + eps;
                const T* src = x + (b * C + c) * (size_t)HW + hw;
                T v[PX];
                if constexpr (PX == 4 && sizeof(T) == 1) {
                    const unsigned w = *reinterpret_cast<const unsigned*>(src);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (T)((w >> (8 * j)) & 0xffu);
                } else if constexpr (PX == 4) {
                    const float4 w = *reinterpret_cast<const float4*>(src);
                    v[0] = (T)w.x; v[1] = (T)w.y; v[2] = (T)w.z; v[3] = (T)w.w;
                } else {
#pragma unroll
                    for (int j = 0; j < PX; ++j) v[j] = src[j];
                }
#pragma unroll
                for (int j = 0; j < PX; ++j) o[j][c] = norm_clip((float)v[j], mu, denom, lo, hi);
            } else {
#pragma unroll
                for (int j = 0; j < PX; ++j) o[j][c] = 0.f;
            }
        }
        float4* dst = reinterpret_cast<float4*>(y + (b * (size_t)HW + hw) * 4);
#pragma unroll
        for (int j = 0; j < PX; ++j) dst[j] = make_float4(o[j][0], o[j][1], o[j][2], o[j][3]);
    }
}

// torch upsample_bilinear2d (align_corners=False): src = scale * (dst + 0.5) - 0.5, clamped at 0;
// i0 = min(int(src), in - 1); i1 = min(i0 + 1, in - 1); l1 = clamp(src - i0, 0, 1); l0 = 1 - l1.
__device__ __forceinline__ void bilinear_tap(int dst, float scale, int in_size, int& i0, int& i1, float& l0, float& l1) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = min((int)src, in_size - 1);
    i1 = min(i0 + 1, in_size - 1);
    l1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
    l0 = 1.f - l1;
}

template <typename T>
__global__ __launch_bounds__(256) void k_resize_bilinear(const T* __restrict__ x, int H1, int W1, int H2, int W2,
                                                         float scale_h, float scale_w, float* __restrict__ y) {
    const int ox = blockIdx.x * blockDim.x + threadIdx.x;
    const int oy = blockIdx.y;
    const int plane = blockIdx.z;
    if (ox >= W2) return;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    bilinear_tap(oy, scale_h, H1, y0, y1, ly0, ly1);
    bilinear_tap(ox, scale_w, W1, x0, x1, lx0, lx1);
    const T* p = x + (size_t)plane * H1 * W1;
    const float v00 = (float)p[(size_t)y0 * W1 + x0];
    const float v01 = (float)p[(size_t)y0 * W1 + x1];
    const float v10 = (float)p[(size_t)y1 * W1 + x0];
    const float v11 = (float)p[(size_t)y1 * W1 + x1];
    const float top = lx0 * v00 + lx1 * v01;
    const float bot = lx0 * v10 + lx1 * v11;
    y[((size_t)plane * H2 + oy) * W2 + ox] = ly0 * top + ly1 * bot;
}

// S == 1: one workgroup per row of E contiguous floats.
__global__ __launch_bounds__(256) void k_l2norm_rows(const float* __restrict__ x, int E, float eps,
                                                     float* __restrict__ y) {
    const float* p = x + (size_t)blockIdx.x * E;
    float* o = y + (size_t)blockIdx.x * E;
    float acc = 0.f;
    for (int i = threadIdx.x; i < E; i += 256) {
        const float v = p[i];
        acc += v * v;
    }
    acc = isc_wave_sum(acc);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    const float denom = fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), eps);
    for (int i = threadIdx.x; i < E; i += 256) o[i] = __fdiv_rn(p[i], denom);
}

// S > 1: x [B, E, S]; a workgroup owns 64 consecutive spatial positions of one image, 4 channel groups.
__global__ __launch_bounds__(256) void k_l2norm_spatial(const float* __restrict__ x, int E, int S, float eps,
                                                        float* __restrict__ y) {
    const int s = blockIdx.x * 64 + (threadIdx.x & 63);
    const int g = threadIdx.x >> 6;
    const size_t base = (size_t)blockIdx.y * E * S;
    float acc = 0.f;
    if (s < S)
        for (int e = g; e < E; e += 4) {
            const float v = x[base + (size_t)e * S + s];
            acc += v * v;
        }
    __shared__ float red[4][64];
    red[g][threadIdx.x & 63] = acc;
    __syncthreads();
    if (s >= S) return;
    const int l = threadIdx.x & 63;
    const float denom = fmaxf(sqrtf(red[0][l] + red[1][l] + red[2][l] + red[3][l]), eps);
    for (int e = g; e < E; e += 4) {
        const size_t i = base + (size_t)e * S + s;
        y[i] = __fdiv_rn(x[i], denom);
    }
}

int grid_for(size_t work_items, int threads, int cap_blocks = 256 * 8) {
    size_t blocks = isc_ceil_div(work_items, (size_t)threads);
    if (blocks > (size_t)cap_blocks) blocks = cap_blocks;
    if (blocks == 0) blocks = 1;
    return (int)blocks;
}

}  // namespace

extern "C" int isc_channel_stats_workspace_bytes(int dtype, int B, int C, int H, int W, size_t* bytes) {
    ISC_REQUIRE(bytes && B > 0 && C > 0 && H > 0 && W > 0);
    ISC_REQUIRE(dtype == ISC_U8 || dtype == ISC_F32);
    const size_t hw = (size_t)H * W;
    const size_t items = (size_t)B * isc_ceil_div(hw, (size_t)kStatsChunk);
    *bytes = isc_align_up((size_t)C * items * sizeof(StatPartial), 256);
    return ISC_OK;
}

extern "C" int isc_channel_stats(const void* x, int dtype, int B, int C, int H, int W, float* mean, float* stdev,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    ISC_REQUIRE(x && mean && stdev);
    size_t need = 0;
    const int st = isc_channel_stats_workspace_bytes(dtype, B, C, H, W, &need);
    if (st != ISC_OK) return st;
    if (!workspace || workspace_bytes < need) return ISC_ERR_WORKSPACE;
    const size_t hw64 = (size_t)H * W;
    if (hw64 > 0x7fffffffu || C > 65535) return ISC_ERR_UNSUPPORTED;
    const int HW = (int)hw64;
    const int chunks = (int)isc_ceil_div(HW, kStatsChunk);
    const size_t items64 = (size_t)B * chunks;
    if (items64 > 0x7fffffffu) return ISC_ERR_UNSUPPORTED;
    const int items = (int)items64;
    StatPartial* part = static_cast<StatPartial*>(workspace);
    const dim3 grid(items, C);
    if (dtype == ISC_U8) {
        const int vec_ok = (HW % 16 == 0) && isc_aligned(x, 16);
        hipLaunchKernelGGL(k_stats_partial_u8, grid, dim3(kStatsThreads), 0, isc_stream(stream),
                           static_cast<const uint8_t*>(x), C, HW, chunks, part, items, vec_ok);
    } else {
        const int vec_ok = (HW % 4 == 0) && isc_aligned(x, 16);
        hipLaunchKernelGGL(k_stats_partial_f32, grid, dim3(kStatsThreads), 0, isc_stream(stream),
                           static_cast<const float*>(x), C, HW, chunks, part, items, vec_ok);
    }
    hipLaunchKernelGGL(k_stats_final, dim3(C), dim3(kStatsThreads), 0, isc_stream(stream), part, items,
                       (double)B * (double)HW, mean, stdev);
    return isc_launch_status();
}

#ifdef ISC_ABLATION
static bool normalize_vec16() {
    static const bool v = getenv("ISC_NORMALIZE_VEC16") != nullptr;  // A/B aid: the 16-pixels-per-lane form, as before
    return v;
}
#else
static constexpr bool normalize_vec16() { return false; }
#endif

extern "C" int isc_normalize_clip(const void* x, int dtype, int B, int C, int H, int W, const float* mean,
                                  const float* stdev, int stat_batch, float eps, float lo, float hi, float* y,
                                  void* stream) {
    ISC_REQUIRE(x && mean && stdev && y && B > 0 && C > 0 && H > 0 && W > 0);
    ISC_REQUIRE(dtype == ISC_U8 || dtype == ISC_F32);
    ISC_REQUIRE(stat_batch == 1 || stat_batch == B);
    const size_t hw64 = (size_t)H * W;
    if (hw64 > 0x7fffffffu) return ISC_ERR_UNSUPPORTED;
    const int HW = (int)hw64;
    const size_t n = (size_t)B * C * HW;
    const int per_image = (stat_batch == B && B > 1) ? 1 : 0;
    hipStream_t s = isc_stream(stream);
    if (dtype == ISC_U8) {
        const uint8_t* p = static_cast<const uint8_t*>(x);
        if (HW % 4 == 0 && isc_aligned(p, 4) && isc_aligned(y, 16) && !normalize_vec16()) {
            const size_t nvec = n / 4;
            hipLaunchKernelGGL(k_normalize_u8_vec4, dim3(grid_for(nvec, 256)), dim3(256), 0, s, p, nvec, C, HW, mean,
                               stdev, per_image, eps, lo, hi, y);
        } else if (HW % 16 == 0 && isc_aligned(p, 16) && isc_aligned(y, 16)) {
            const size_t nvec = n / 16;
            hipLaunchKernelGGL(k_normalize_u8_vec16, dim3(grid_for(nvec, 256)), dim3(256), 0, s, p, nvec, C, HW, mean,
                               stdev, per_image, eps, lo, hi, y);
        } else {
            hipLaunchKernelGGL(k_normalize_scalar<uint8_t>, dim3(grid_for(n, 256)), dim3(256), 0, s, p, n, C, HW, mean,
                               stdev, per_image, eps, lo, hi, y);
        }
    } else {
        const float* p = static_cast<const float*>(x);
        if (HW % 4 == 0 && isc_aligned(p, 16) && isc_aligned(y, 16)) {
            const size_t nvec = n / 4;
            hipLaunchKernelGGL(k_normalize_f32_vec4, dim3(grid_for(nvec, 256)), dim3(256), 0, s, p, nvec, C, HW, mean,
                               stdev, per_image, eps, lo, hi, y);
        } else {
            hipLaunchKernelGGL(k_normalize_scalar<float>, dim3(grid_for(n, 256)), dim3(256), 0, s, p, n, C, HW, mean,
                               stdev, per_image, eps, lo, hi, y);
        }
    }
    return isc_launch_status();
}

#ifdef ISC_ABLATION
static bool normalize_no_transpose() {
    static const bool v = getenv("ISC_NORMALIZE_NO_TRANSPOSE") != nullptr;  // A/B aid: strided 16-byte stores, as before
    return v;
}
#else
static constexpr bool normalize_no_transpose() { return false; }
#endif

// The four-pixels-per-thread form with the stores TRANSPOSED through LDS: a thread still reads its four consecutive pixels of
// each plane with one load, but a store instruction of k_normalize_nhwc4<T, 4> writes 64 pieces of 16 bytes that lie 64 bytes
// apart (every 128-byte line is written by four instructions), and the float32 output is 5/6 of this kernel's traffic.  Here
// the 1 024 pixels of a workgroup pass through 16 KiB of LDS and thread t stores pixels t, t + 256, ...: a wave writes one
// contiguous KiB per instruction.  Same arithmetic, bit-identical output.
template <typename T>
__global__ __launch_bounds__(256) void k_normalize_nhwc4_t(const T* __restrict__ x, size_t ngroups, int C, int HW,
                                                           const float* __restrict__ mean, const float* __restrict__ stdev,
                                                           int per_image, float eps, float lo, float hi,
                                                           float* __restrict__ y) {
    __shared__ float4 px[1024];
    const int tid = threadIdx.x;
    const size_t groups_per_image = (size_t)HW / 4;
    for (size_t base = (size_t)blockIdx.x * 256; base < ngroups; base += (size_t)gridDim.x * 256) {
        const size_t g = base + tid;
        if (g < ngroups) {
            const size_t b = g / groups_per_image;
            const size_t hw = (g - b * groups_per_image) * 4;
            float o[4][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c < C) {
                    const int st = per_image ? (int)(b * C + c) : c;
                    const float mu = mean[st];
                    const float denom = stdev[st] + eps;
                    const T* src = x + (b * C + c) * (size_t)HW + hw;
                    T v[4];
                    if constexpr (sizeof(T) == 1) {
                        const unsigned w = *reinterpret_cast<const unsigned*>(src);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = (T)((w >> (8 * j)) & 0xffu);
                    } else {
                        const float4 w = *reinterpret_cast<const float4*>(src);
                        v[0] = (T)w.x; v[1] = (T)w.y; v[2] = (T)w.z; v[3] = (T)w.w;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j][c] = norm_clip((float)v[j], mu, denom, lo, hi);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j][c] = 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) px[tid * 4 + j] = make_float4(o[j][0], o[j][1], o[j][2], o[j][3]);
        }
        __syncthreads();
        // y is [B][HW][4]: the workgroup's 1 024 pixels are consecutive in it whatever image they belong to
        const size_t npx = (ngroups - base < 256 ? ngroups - base : 256) * 4;
        float4* dst = reinterpret_cast<float4*>(y) + base * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((size_t)(j * 256 + tid) < npx) dst[j * 256 + tid] = px[j * 256 + tid];
        __syncthreads();
    }
}

extern "C" int isc_normalize_clip_nhwc4(const void* x, int dtype, int B, int C, int H, int W, const float* mean,
                                        const float* stdev, int stat_batch, float eps, float lo, float hi, float* y,
                                        void* stream) {
    ISC_REQUIRE(x && mean && stdev && y && B > 0 && C > 0 && C <= 4 && H > 0 && W > 0);
    ISC_REQUIRE(dtype == ISC_U8 || dtype == ISC_F32);
    ISC_REQUIRE(stat_batch == 1 || stat_batch == B);
    const size_t hw64 = (size_t)H * W;
    if (hw64 > 0x7fffffffu) return ISC_ERR_UNSUPPORTED;
    if (!isc_aligned(y, 16)) return ISC_ERR_ALIGNMENT;
    const int HW = (int)hw64;
    const int per_image = (stat_batch == B && B > 1) ? 1 : 0;
    hipStream_t s = isc_stream(stream);
    const bool vec = HW % 4 == 0 && isc_aligned(x, 16);
    const size_t ngroups = (size_t)B * (vec ? HW / 4 : HW);
    const dim3 grid(grid_for(ngroups, 256)), block(256);
    if (dtype == ISC_U8) {
        const uint8_t* p = static_cast<const uint8_t*>(x);
        if (vec && !normalize_no_transpose()) hipLaunchKernelGGL((k_normalize_nhwc4_t<uint8_t>), grid, block, 0, s, p, ngroups, C, HW, mean, stdev, per_image, eps, lo, hi, y);
        else if (vec) hipLaunchKernelGGL((k_normalize_nhwc4<uint8_t, 4>), grid, block, 0, s, p, ngroups, C, HW, mean, stdev, per_image, eps, lo, hi, y);
        else hipLaunchKernelGGL((k_normalize_nhwc4<uint8_t, 1>), grid, block, 0, s, p, ngroups, C, HW, mean, stdev, per_image, eps, lo, hi, y);
    } else {
        const float* p = static_cast<const float*>(x);
        if (vec && !normalize_no_transpose()) hipLaunchKernelGGL((k_normalize_nhwc4_t<float>), grid, block, 0, s, p, ngroups, C, HW, mean, stdev, per_image, eps, lo, hi, y);
        else if (vec) hipLaunchKernelGGL((k_normalize_nhwc4<float, 4>), grid, block, 0, s, p, ngroups, C, HW, mean, stdev, per_image, eps, lo, hi, y);
        else hipLaunchKernelGGL((k_normalize_nhwc4<float, 1>), grid, block, 0, s, p, ngroups, C, HW, mean, stdev, per_image, eps, lo, hi, y);
    }
    return isc_launch_status();
}

extern "C" int isc_resize_bilinear(const void* x, int dtype, int planes, int H1, int W1, int H2, int W2, float* y,
                                   void* stream) {
    ISC_REQUIRE(x && y && planes > 0 && H1 > 0 && W1 > 0 && H2 > 0 && W2 > 0);
    ISC_REQUIRE(dtype == ISC_U8 || dtype == ISC_F32);
    if (H2 > 65535 || planes > 65535) return ISC_ERR_UNSUPPORTED;
    const float scale_h = (float)H1 / (float)H2;
    const float scale_w = (float)W1 / (float)W2;
    const dim3 grid(isc_ceil_div(W2, 256), H2, planes);
    if (dtype == ISC_U8)
        hipLaunchKernelGGL(k_resize_bilinear<uint8_t>, grid, dim3(256), 0, isc_stream(stream),
                           static_cast<const uint8_t*>(x), H1, W1, H2, W2, scale_h, scale_w, y);
    else
        hipLaunchKernelGGL(k_resize_bilinear<float>, grid, dim3(256), 0, isc_stream(stream),
                           static_cast<const float*>(x), H1, W1, H2, W2, scale_h, scale_w, y);
    return isc_launch_status();
}

extern "C" int isc_l2norm_channels(const float* x, int B, int E, int S, float eps, float* y, void* stream) {
    ISC_REQUIRE(x && y && B > 0 && E > 0 && S > 0);
    if (S == 1) {
        hipLaunchKernelGGL(k_l2norm_rows, dim3(B), dim3(256), 0, isc_stream(stream), x, E, eps, y);
    } else {
        if (B > 65535) return ISC_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(k_l2norm_spatial, dim3(isc_ceil_div(S, 64), B), dim3(256), 0, isc_stream(stream), x, E, S,
                           eps, y);
    }
    return isc_launch_status();
}
