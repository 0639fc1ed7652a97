// Microbenchmark: do vector stores overlap with MFMA issue on MI355X, and what does a store cost the issuing wave?
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/store_wall scripts/microbench/store_wall.hip && /tmp/store_wall
// One 512-thread workgroup per CU; per "tile" a wave issues 768 MFMAs (16x16x32 f16, operands in registers) and 16
// dwordx4 stores in the layout of k_gemm_f16_stream's epilogue (16 tokens x 4 lanes x 32 B per store pair).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#pragma clang diagnostic ignored "-Wunused-value"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: MFMA only   1: stores only   2: MFMAs then the 16 stores   3: one store after every 48 MFMAs
//      4: the 16 stores inside the first 96 MFMAs   5: like 3 but 4-byte stores
//      6: like 3, but only waves 0 - 3 store (two stores after every 48 MFMAs); waves 4 - 7 issue MFMAs only
//      7: waves 0 - 3 store (two per 48 MFMAs) and issue HALF the MFMAs; waves 4 - 7 issue 1.5 x the MFMAs
template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned char* out, int tiles, int n_feat_bytes, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, fg = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (lane + j)); b[j] = (_Float16)(0.002f * (lane - j)); }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
    for (int t = 0; t < tiles; ++t) {
        const size_t tile_row0 = ((size_t)blockIdx.x * tiles + t) * 256 + wm * 128;
        auto store = [&](int s) {  // s = 0..15: row block s >> 1, half s & 1
            const size_t token = tile_row0 + (s >> 1) * 16 + frow;
            unsigned char* dst = out + token * n_feat_bytes + (wn * 64 + fg * 16) * 2 + (s & 1) * 16;
            if (MODE == 5) *reinterpret_cast<unsigned*>(dst) = v.x + s;
            else *reinterpret_cast<uint4*>(dst) = make_uint4(v.x + s, v.y, v.z, v.w);
        };
        if (MODE == 1) {
            for (int s = 0; s < 16; ++s) store(s);
            continue;
        }
        const bool storer = __builtin_amdgcn_readfirstlane(wave) < 4;
        if (MODE == 7 && !storer) {  // the partner wave of each SIMD: 1.5 x the matrix work, no memory instructions
#pragma unroll 1
            for (int blk = 0; blk < 24; ++blk) {
#pragma unroll
                for (int j = 0; j < 48; ++j) acc[j & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j & 7], 0, 0, 0);
            }
            continue;
        }
#pragma unroll 1
        for (int blk = 0; blk < (MODE == 7 ? 8 : 16); ++blk) {  // 16 blocks of 48 MFMAs = one tile's 768
#pragma unroll
            for (int j = 0; j < 48; ++j) {
                acc[j & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j & 7], 0, 0, 0);
                if (MODE == 4 && blk < 2 && (j % 6) == 5) store(blk * 8 + j / 6);  // 16 stores inside the first 96 MFMAs
            }
            if (MODE == 3 || MODE == 5) store(blk);
            if (MODE == 6 && storer) { store(blk); store(blk ^ 1); }
            if (MODE == 7) { store(2 * blk); store(2 * blk + 1); store((2 * blk) ^ 8); store((2 * blk + 1) ^ 8); }
        }
        if (MODE == 2) for (int s = 0; s < 16; ++s) store(s);
    }
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3];
    if (r == 12345.678f) sink[0] = r;
}

template <int MODE>
float run(unsigned char* out, int tiles, int nfb, float* sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, tiles, nfb, sink);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, tiles, nfb, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    const int tiles = 15, nfb = 256 * 2;  // each workgroup writes its own 256 x 256 fp16 tiles, 512-byte rows
    const size_t bytes = (size_t)256 * tiles * 256 * nfb;
    unsigned char* out; float* sink;
    hipMalloc(&out, bytes); hipMalloc(&sink, 4);
    const double mb = bytes / 1e6, flops = 2.0 * 256 * tiles * 8 * 768 * 16 * 16 * 32;
    const char* names[] = {"MFMA only", "stores only", "MFMAs then 16 stores", "1 store per 48 MFMAs", "16 stores inside the first 96 MFMAs", "1 dword store per 48 MFMAs",
                           "waves 0-3: 2 stores per 48 MFMAs; 4-7: none", "waves 0-3: stores + 1/2 MFMAs; 4-7: 3/2 MFMAs"};
    float t[8] = {run<0>(out, tiles, nfb, sink), run<1>(out, tiles, nfb, sink), run<2>(out, tiles, nfb, sink),
                  run<3>(out, tiles, nfb, sink), run<4>(out, tiles, nfb, sink), run<5>(out, tiles, nfb, sink),
                  run<6>(out, tiles, nfb, sink), run<7>(out, tiles, nfb, sink)};
    for (int m = 0; m < 8; ++m)
        printf("%-40s %8.3f ms   %7.1f TFLOP/s   %7.1f GB/s written\n", names[m], t[m], m == 1 ? 0.0 : flops / t[m] / 1e9,
               m == 0 ? 0.0 : (m == 5 ? mb / 4 : mb) / t[m]);
    printf("(%.0f MB of output per launch; MFMA+store perfectly overlapped = max of rows 0 and 1)\n", mb);
    return 0;
}
