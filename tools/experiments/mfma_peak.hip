// Practical MFMA ceiling on this chip: back-to-back v_mfma_f32_32x32x16_f16 from registers, random operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(const half8* in, float* out, int iters) {
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x + 256 * i]; b[i] = in[threadIdx.x + 256 * (i + 4)]; }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 3], b[(u >> 2) & 3], acc[u % NACC], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    half8* d; float* o; const int n = 256 * 8;
    hipMalloc(&d, n * sizeof(half8)); hipMalloc(&o, 4096 * 256 * 4);
    _Float16* h = (_Float16*)malloc(n * 16);
    for (int i = 0; i < n * 8; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
    hipMemcpy(d, h, n * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int blocks : {256, 512, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double fl = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
            printf("acc4 blocks %d (%.1f waves/SIMD): %.3f ms  %.0f TFLOP/s\n", blocks, blocks / 256.0, ms, fl / ms / 1e9);
        }
    }
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<1>, dim3(512), dim3(256), 0, 0, d, o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("acc1 blocks 512: %.3f ms  %.0f TFLOP/s\n", ms, (double)512 * 4 * iters * 16 * 2.0 * 32 * 32 * 16 / ms / 1e9);
    return 0;
}
