// LDS bank-conflict probe: run under  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
// Each variant = a distinct kernel name; every lane of a 256-thread block issues ITERS ds_read_b128
// (or ds_write_b128) at a per-lane address defined by the variant.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
#define ITERS 512

template <int V>
__device__ int addr_of(int tid) {
    const int lane = tid & 63, r = lane & 31, hh = lane >> 5, wave = tid >> 6;
    const int c = 2 * (wave & 3) + hh;  // some chunk
    switch (V) {
        case 0: return r * 128 + c * 16;                                    // linear
        case 1: return r * 128 + ((c ^ ((r >> 1) & 7)) << 4);               // v1 swizzle
        case 2: return r * 128 + ((c ^ (r & 7)) << 4);                      // row&7
        case 3: return r * 144 + c * 16;                                    // padded pitch 144
        case 4: return r * 128 + ((c ^ ((r >> 2) & 7)) << 4);               // row>>2
        case 5: return (r >> 1) * 256 + ((((r & 1) * 8 + c) ^ ((r >> 1) & 15)) << 4);   // 256-B superrow, xor row&15
        case 6: return lane * 16;                                           // fully linear 16B per lane (ideal)
        case 7: return r * 272 + c * 16;                                    // pitch 272
        // writes (staging pattern): chunk = tid&7, row = tid>>3
        case 10: { int ch = tid & 7, row = tid >> 3; return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }
        case 11: { int ch = tid & 7, row = tid >> 3; return row * 128 + ch * 16; }
        case 12: { int ch = tid & 7, row = tid >> 3; return row * 144 + ch * 16; }
        case 13: return tid * 16;
        case 14: { int ch = tid & 7, row = tid >> 3; return row * 128 + ((ch ^ (row & 7)) << 4); }
        case 15: { int row = tid & 31, ch = tid >> 5; return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }   // lanes walk rows
        case 16: { int row = tid & 31, ch = tid >> 5; return row * 144 + ch * 16; }
        // 64-byte rows, 16x16x32 fragments (lane = l16 + 16 kq: row / 4x4-block pixel l16, 16-byte chunk kq) — conv3x3_pw
        case 20: { int l16 = lane & 15, kq = lane >> 4; return l16 * 64 + (((kq + (l16 >> 2)) & 3) << 4); }                  // weights, round-2 rotation
        case 21: { int l16 = lane & 15, kq = lane >> 4; return l16 * 64 + ((kq ^ (((l16 >> 2) & 1) << 1)) << 4); }           // weights, xor 2*(row>>2 & 1)
        case 22: { int l16 = lane & 15, kq = lane >> 4; return ((l16 >> 2) * 12 + (l16 & 3) + 13) * 64 + (((kq + (l16 >> 2) + 1) & 3) << 4); }   // patch tap (1,1), rotation
        case 23: { int l16 = lane & 15, kq = lane >> 4; int y = (l16 >> 2) + 1; return (y * 12 + (l16 & 3) + 1) * 64 + ((kq ^ ((y & 1) << 1)) << 4); }   // patch tap (1,1), xor 2*(y & 1)
        case 24: { int l16 = lane & 15, kq = lane >> 4; int y = (l16 >> 2) + 1; return (y * 9 + (l16 & 3) + 1) * 64 + ((kq ^ ((y & 1) << 1)) << 4); }    // the same on an odd pitch (9 cells)
    }
    return 0;
}

template <int V, bool WRITE>
__global__ __launch_bounds__(256) void probe(float* out) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    const int a = addr_of<V>(threadIdx.x);
    f4 acc = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) ((float*)smem)[i] = (float)i;
    __syncthreads();
    for (int it = 0; it < ITERS; ++it) {
        const int off = (it & 1) * 32768;  // keeps the compiler from hoisting
        if (WRITE) {
            acc[0] += 1.f;
            asm volatile("ds_write_b128 %0, %1" :: "v"(off + a), "v"(acc) : "memory");
        } else {
            f4 v = *(volatile f4*)(smem + off + a);
            acc += v;
        }
    }
    __syncthreads();
    if (WRITE) acc = *(f4*)(smem + (threadIdx.x & 63) * 16);
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
    float* d;
    (void)hipMalloc(&d, 1024 * 256 * 4);
#define RUN(V, W) hipLaunchKernelGGL((probe<V, W>), dim3(256), dim3(256), 0, 0, d);
    RUN(0, false) RUN(1, false) RUN(2, false) RUN(3, false) RUN(4, false) RUN(5, false) RUN(6, false) RUN(7, false)
    RUN(20, false) RUN(21, false) RUN(22, false) RUN(23, false) RUN(24, false)
    RUN(10, true) RUN(11, true) RUN(12, true) RUN(13, true) RUN(14, true) RUN(15, true) RUN(16, true)
    (void)hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
