// PCA fit on the device (include/imagescry_hip.h: isc_feature_sums, isc_center_transpose; the Gram matrix itself is
// isc_gram_rows in encoder.hip, on the f32 matrix cores).
//
// The reference centres [N, F] rows and takes their SVD on the host (src/imagescry/models/decomposition.py:118-131).  Here
// the N-sized work stays on the GPU: per-feature sums in float64 (two deterministic stages), then chunk by chunk the
// centred rows are written TRANSPOSED ([F, chunk]: the sample axis becomes the contiguous reduction axis both operands of
// the implicit-GEMM kernel want) and the F x F Gram matrix of the chunk is one k_conv_f32 launch.  Only the F x F
// eigenproblem goes to the host (imagescry_amd/decomposition.py).
#include "isc_common.h"

namespace {

constexpr int SUM_ROWS = 1024;  // rows per partial sum

// partial[blk][f] = sum over the block's rows of x[row][f], float64; thread = one feature (coalesced across a row)
__global__ __launch_bounds__(256) void k_feature_sums_partial(const float* __restrict__ x, int64_t n, int F, int64_t ldx,
                                                              double* __restrict__ partial) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    const int64_t r0 = (int64_t)blockIdx.y * SUM_ROWS;
    const int64_t r1 = r0 + SUM_ROWS < n ? r0 + SUM_ROWS : n;
    double acc = 0.0;
    const float* p = x + r0 * ldx + f;
    int64_t r = r0;
    for (; r + 4 <= r1; r += 4) {  // four independent loads in flight
        const float a = p[0], b = p[ldx], c = p[2 * ldx], d = p[3 * ldx];
        acc += (double)a;
        acc += (double)b;
        acc += (double)c;
        acc += (double)d;
        p += 4 * ldx;
    }
    for (; r < r1; ++r, p += ldx) acc += (double)*p;
    partial[(size_t)blockIdx.y * F + f] = acc;
}

// sums[f] = partial[0][f] + partial[1][f] + ... in block order (bit-reproducible)
__global__ __launch_bounds__(256) void k_feature_sums_final(const double* __restrict__ partial, int nblk, int F,
                                                            double* __restrict__ sums) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    double acc = 0.0;
    for (int b = 0; b < nblk; ++b) acc += partial[(size_t)b * F + f];
    sums[f] = acc;
}

// xt[f][i] = x[i][f] - mean[f] for i < n, 0 for n <= i < ldn; rows f >= F (up to Fpad) are zero.  32 x 32 tiles through
// LDS so that both the reads (along f) and the writes (along i) are coalesced.
__global__ __launch_bounds__(256) void k_center_transpose(const float* __restrict__ x, int64_t n, int F, int64_t ldx,
                                                          const float* __restrict__ mean, float* __restrict__ xt,
                                                          int Fpad, int64_t ldn) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int64_t i0 = (int64_t)blockIdx.x * 32;
    const int f0 = blockIdx.y * 32;
    const int f = f0 + tx;
    const float mu = f < F ? mean[f] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = i0 + ty + 8 * j;
        tile[ty + 8 * j][tx] = (i < n && f < F) ? x[i * ldx + f] - mu : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int fo = f0 + ty + 8 * j;
        const int64_t i = i0 + tx;
        if (fo < Fpad && i < ldn) xt[(size_t)fo * ldn + i] = tile[tx][ty + 8 * j];
    }
}

}  // namespace

extern "C" int isc_feature_sums_workspace_bytes(int64_t n, int F, size_t* bytes) {
    ISC_REQUIRE(bytes && n > 0 && F > 0);
    *bytes = (size_t)isc_ceil_div<int64_t>(n, SUM_ROWS) * F * sizeof(double);
    return ISC_OK;
}

extern "C" int isc_feature_sums(const float* x, int64_t n, int F, int64_t ldx, double* sums, void* workspace,
                                size_t workspace_bytes, void* stream) {
    ISC_REQUIRE(x && sums && n > 0 && F > 0 && ldx >= F);
    const int64_t nblk = isc_ceil_div<int64_t>(n, SUM_ROWS);
    if (nblk > 65535) return ISC_ERR_UNSUPPORTED;  // 67 M rows per call: chunk the rows above that
    if (!workspace || workspace_bytes < (size_t)nblk * F * sizeof(double)) return ISC_ERR_WORKSPACE;
    if (!isc_aligned(workspace, 8)) return ISC_ERR_ALIGNMENT;
    hipStream_t s = isc_stream(stream);
    double* partial = static_cast<double*>(workspace);
    hipLaunchKernelGGL(k_feature_sums_partial, dim3(isc_ceil_div(F, 256), (unsigned)nblk), dim3(256), 0, s, x, n, F, ldx,
                       partial);
    hipLaunchKernelGGL(k_feature_sums_final, dim3(isc_ceil_div(F, 256)), dim3(256), 0, s, partial, (int)nblk, F, sums);
    return isc_launch_status();
}

extern "C" int isc_center_transpose(const float* x, int64_t n, int F, int64_t ldx, const float* mean, float* xt, int Fpad,
                                    int64_t ldn, void* stream) {
    ISC_REQUIRE(x && mean && xt && n > 0 && F > 0 && ldx >= F && Fpad >= F && ldn >= n);
    const int64_t gx = isc_ceil_div<int64_t>(ldn, 32);
    if (gx > 0x7fffffff || isc_ceil_div(Fpad, 32) > 65535) return ISC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_center_transpose, dim3((unsigned)gx, isc_ceil_div(Fpad, 32)), dim3(256), 0, isc_stream(stream), x,
                       n, F, ldx, mean, xt, Fpad, ldn);
    return isc_launch_status();
}
