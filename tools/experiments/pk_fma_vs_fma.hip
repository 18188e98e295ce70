// Does v_pk_fma_f32 give the bits of v_fma_f32?  (Round 4: with the ReLU hoisted out of the epilogue's element loop hipcc paired the
// BN fmas of the no-ReLU branch into v_pk_fma_f32, and one output in 10^6 of a 1x1 conv came out one fp16 ulp away from the kernel
// that still used v_fma_f32.)   hipcc --offload-arch=gfx950 -O2 -o tools/experiments/pk_fma_vs_fma tools/experiments/pk_fma_vs_fma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>

typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void k(const float* a, const float* b, const float* c, float* r_fma, float* r_pk, int n) {
    const int i = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (i + 1 >= n) return;
    float x0 = __builtin_fmaf(a[i], b[i], c[i]), x1 = __builtin_fmaf(a[i + 1], b[i + 1], c[i + 1]);
    asm volatile("" : "+v"(x0), "+v"(x1));
    f2 av = {a[i], a[i + 1]}, bv = {b[i], b[i + 1]}, cv = {c[i], c[i + 1]}, d;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(av), "v"(bv), "v"(cv));
    r_fma[i] = x0; r_fma[i + 1] = x1;
    r_pk[i] = d[0]; r_pk[i + 1] = d[1];
}

int main() {
    const int n = 1 << 24;
    std::vector<float> a(n), b(n), c(n), r1(n), r2(n);
    std::mt19937 g(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (int i = 0; i < n; ++i) { a[i] = nd(g) * 8.f; b[i] = 0.5f + 0.5f * (float)(g() & 0xffff) / 65536.f; c[i] = nd(g) * 0.1f; }
    float *da, *db, *dc, *d1, *d2;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&d2, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, da, db, dc, d1, d2, n);
    hipMemcpy(r1.data(), d1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), d2, n * 4, hipMemcpyDeviceToHost);
    long diff = 0, diff_lo = 0, vs_host = 0, unfused = 0;
    for (int i = 0; i < n; ++i) {
        uint32_t u1, u2; memcpy(&u1, &r1[i], 4); memcpy(&u2, &r2[i], 4);
        const float h = __builtin_fmaf(a[i], b[i], c[i]);
        const float prod = a[i] * b[i];
        volatile float un = prod + c[i];
        if (memcmp(&h, &r1[i], 4)) ++vs_host;
        if (u1 != u2) {
            ++diff; if (!(i & 1)) ++diff_lo;
            if (!memcmp((const void*)&un, &r2[i], 4)) ++unfused;
            if (diff <= 5) printf("  a=%a b=%a c=%a  v_fma=%a  v_pk_fma=%a  (half %d)\n", a[i], b[i], c[i], r1[i], r2[i], i & 1);
        }
    }
    printf("%d triples: v_fma_f32 != host fmaf in %ld; v_pk_fma_f32 != v_fma_f32 in %ld (%ld in the low half); of those equal to the UNFUSED a*b+c: %ld\n", n, vs_host, diff, diff_lo, unfused);
    return 0;
}
