// Microbenchmark: what does the index burst of a 1024-thread workgroup cost as 4-byte loads vs 16-byte loads?
// One workgroup per CU (LDS request forces it), every thread loads WORDS dwords of an L2-resident array either as WORDS
// dword loads (lane-contiguous, slot-strided: v = tid + k*1024) or as WORDS/4 dwordx4 loads (tile layout [k/4][tid][4]),
// then the workgroup barriers; reported: cycles from the first load to the barrier exit (s_memtime of wave 0), median over
// blocks.  hipcc --offload-arch=gfx950 -O3 -o load_width load_width.hip && ./load_width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
constexpr int WG = 1024, WORDS = 40;
template <int MODE>
__global__ __launch_bounds__(WG) void k(const uint32_t *__restrict__ a, unsigned long long *t, uint32_t *sink) {
    extern __shared__ unsigned char smem[];
    const int tid = threadIdx.x;
    const uint32_t *base = a + (size_t)blockIdx.x * WORDS * WG;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    uint32_t acc = 0;
    if (MODE == 0) {
        uint32_t w[WORDS];
#pragma unroll
        for (int i = 0; i < WORDS; i++) w[i] = base[i * WG + tid];
#pragma unroll
        for (int i = 0; i < WORDS; i++) acc ^= w[i];
    } else if (MODE == 1) {
        uint4 w[WORDS / 4];
        const uint4 *b4 = reinterpret_cast<const uint4 *>(base);
#pragma unroll
        for (int i = 0; i < WORDS / 4; i++) w[i] = b4[i * WG + tid];
#pragma unroll
        for (int i = 0; i < WORDS / 4; i++) acc ^= w[i].x ^ w[i].y ^ w[i].z ^ w[i].w;
    } else {
        uint2 w[WORDS / 2];
        const uint2 *b2 = reinterpret_cast<const uint2 *>(base);
#pragma unroll
        for (int i = 0; i < WORDS / 2; i++) w[i] = b2[i * WG + tid];
#pragma unroll
        for (int i = 0; i < WORDS / 2; i++) acc ^= w[i].x ^ w[i].y;
    }
    reinterpret_cast<uint32_t *>(smem)[tid] = acc;
    __syncthreads();
    const unsigned long long t1 = wall_clock64();
    if (tid == 0) t[blockIdx.x] = t1 - t0;
    if (acc == 0x12345678u) sink[0] = acc;
}
int main() {
    const int NB = 176;
    uint32_t *a, *sink; unsigned long long *t;
    hipMalloc(&a, sizeof(uint32_t) * (size_t)NB * WORDS * WG);
    hipMemset(a, 1, sizeof(uint32_t) * (size_t)NB * WORDS * WG);
    hipMalloc(&t, sizeof(unsigned long long) * NB); hipMalloc(&sink, 4);
    std::vector<unsigned long long> h(NB);
    const char *names[3] = {"dword x40", "dwordx4 x10", "dwordx2 x20"};
    for (int rep = 0; rep < 3; rep++)
        for (int mode = 0; mode < 3; mode++) {
            hipFuncSetAttribute((const void *)(mode == 0 ? k<0> : mode == 1 ? k<1> : k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(NB), dim3(WG), 100 * 1024, 0, a, t, sink);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(NB), dim3(WG), 100 * 1024, 0, a, t, sink);
            else hipLaunchKernelGGL(k<2>, dim3(NB), dim3(WG), 100 * 1024, 0, a, t, sink);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), t, sizeof(unsigned long long) * NB, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            printf("rep %d %-12s: median %.2f us  min %.2f  max %.2f  (100 MHz wall clock; %d B per thread)\n", rep, names[mode],
                   h[NB / 2] / 100.0, h[0] / 100.0, h[NB - 1] / 100.0, WORDS * 4);
        }
    return 0;
}
