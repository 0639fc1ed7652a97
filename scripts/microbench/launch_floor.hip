// Launch-floor microbenchmark: what does an early-exit kernel cost on the stream, by grid / block / dynamic LDS?
// hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip && ./launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

__global__ void k_gate(const int* flag, int* out) {
    extern __shared__ unsigned char dyn[];
    if (*flag <= 0) return;
    out[blockIdx.x] = dyn[threadIdx.x];
}
__global__ void k_touch(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[1] += 1; }
// keeps the GPU busy for ~`us` microseconds so that the launches behind it are all queued before they can start
__global__ void k_spin(int* p, long long us) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < us * 100) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) p[2] += 1;
}

int main() {
    int* d;
    hipMalloc(&d, 1 << 20);
    hipMemset(d, 0, 1 << 20);
    hipStream_t s;
    hipStreamCreate(&s);
    hipFuncSetAttribute((const void*)k_gate, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    struct Cfg { int grid, block, lds; };
    const Cfg cfgs[] = {{1, 64, 0}, {1, 512, 0}, {16, 512, 0}, {64, 64, 0}, {64, 256, 0}, {64, 512, 0}, {256, 64, 0}, {256, 256, 0},
                        {256, 512, 0}, {256, 512, 64 * 1024}, {256, 512, 150 * 1024}, {64, 512, 150 * 1024}, {1024, 512, 0}, {1024, 64, 0}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int REP = 200;
    for (int queued = 0; queued < 2; ++queued)
    for (const Cfg& c : cfgs) {
        std::vector<float> ts;
        for (int trial = 0; trial < 5; ++trial) {
            // chain: touch (a real predecessor), then REP gated kernels back to back; report per gated kernel
            // QUEUED = 1: a 3 ms spinner first, so the host is far ahead and the gaps are the GPU's own
            if (queued) hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, d, 3000LL);
            else hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, s, d);
            hipEventRecord(e0, s);
            for (int i = 0; i < REP; ++i) hipLaunchKernelGGL(k_gate, dim3(c.grid), dim3(c.block), c.lds, s, d, d + 1024);
            hipEventRecord(e1, s);
            hipStreamSynchronize(s);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            ts.push_back(ms * 1e3f / REP);
        }
        std::sort(ts.begin(), ts.end());
        printf("%s grid %5d block %4d lds %6d KiB : %.2f us per early-exit launch (median of 5, %d back to back)\n",
               queued ? "queued " : "host-paced", c.grid, c.block, c.lds / 1024, ts[2], REP);
    }
    return 0;
}
