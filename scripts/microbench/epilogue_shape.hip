// Epilogue access-shape microbenchmark: out = max(res + 1, 0) over an [M][C] float32 tensor, 128 x 128 tiles walked by two
// persistent 256-thread workgroups per CU, 16 x 16-byte loads then 16 x 16-byte stores per lane and tile -- once in the
// shape a 16x16 MFMA accumulator gives k_conv_f32's epilogue (a wave instruction = 16 pixel rows x 64 contiguous bytes),
// once row-contiguous (a wave instruction = 2 pixel rows x 512 contiguous bytes).  Is the accumulator shape what keeps the
// short-K 1 x 1 + residual layers (ResNet-50 l1.*.conv3: 4.5 TB/s) off the streaming rate?
// hipcc --offload-arch=gfx950 -O3 -o epilogue_shape epilogue_shape.hip && ./epilogue_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void k_epi(const float* __restrict__ res, float* __restrict__ out, int M, int C,
                                               int ntiles) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_co = C / 128;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int co0 = (tile % n_co) * 128, pix0 = (tile / n_co) * 128;
        f32x4 v[16];
        size_t off[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int m, co;
            if (SHAPE == 0) {  // accumulator shape: wave = (64 channels, 64 pixels); i = ni * 4 + mi
                const int wco = wave & 1, wpix = wave >> 1, ni = i >> 2, mi = i & 3;
                m = pix0 + wpix * 64 + ni * 16 + (lane & 15);
                co = co0 + wco * 64 + mi * 16 + (lane >> 4) * 4;
            } else {  // row shape: wave = 32 pixels x 128 channels; an instruction = 2 rows x 512 B
                m = pix0 + wave * 32 + i * 2 + (lane >> 5);
                co = co0 + (lane & 31) * 4;
            }
            off[i] = (size_t)min(m, M - 1) * C + co;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const f32x4*>(res + off[i]);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            f32x4 r = v[i] + 1.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = fmaxf(r[e], 0.f);
            *reinterpret_cast<f32x4*>(out + off[i]) = r;
        }
    }
}

int main() {
    const int M = 512 * 56 * 56;
    for (int C : {256, 512, 128}) {
        const size_t n = (size_t)M * C;
        float *res, *out;
        hipMalloc(&res, n * 4);
        hipMalloc(&out, n * 4);
        hipMemset(res, 0, n * 4);
        const int ntiles = (M / 128) * (C / 128);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int shape = 0; shape < 2; ++shape) {
            std::vector<float> ts;
            for (int trial = 0; trial < 5; ++trial) {
                hipEventRecord(e0, 0);
                for (int r = 0; r < 3; ++r) {
                    if (shape == 0) hipLaunchKernelGGL(k_epi<0>, dim3(512), dim3(256), 0, 0, res, out, M, C, ntiles);
                    else hipLaunchKernelGGL(k_epi<1>, dim3(512), dim3(256), 0, 0, res, out, M, C, ntiles);
                }
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                ts.push_back(ms / 3);
            }
            std::sort(ts.begin(), ts.end());
            printf("C = %4d  %s: %.1f us  %.2f TB/s (read + write)\n", C,
                   shape == 0 ? "accumulator shape (16 rows x 64 B)" : "row shape         (2 rows x 512 B)", ts[2] * 1e3,
                   2.0 * n * 4 / (ts[2] * 1e-3) / 1e12);
        }
        hipFree(res);
        hipFree(out);
    }
    return 0;
}
