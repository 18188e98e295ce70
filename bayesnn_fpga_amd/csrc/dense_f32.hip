// Hidden fully-connected layers on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 FMA chain).
// K is split between the two lane halves (lane half h takes k in [h*K/2, (h+1)*K/2)), which a dot product does not care
// about, so every lane streams its row with float4 loads.  (The exit classifier itself lives in head_fused.hip.)
#include "conv_epilogue.h"
#include "kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------------------
// Hidden fully-connected layer of a classifier stack (VGG-11's Dense 512 / 512, Hardware_Artifact/bayes_hw/models/
// models.py:262-281), kept in fp32 end to end: out[n][c] = relu?(in[n % in_mod] . W[c] + bias[c]) (site).
// With fp16 outputs (the layer as a 1x1 conv) two more roundings land on logits of magnitude ~10 and the predictive mean
// misses the 1e-3 bar at B = 250 (measured 1.3e-3); the layer is 0.1 % of the FLOPs, so it runs on the exact-f32 MFMA
// like the classifier above.  One wave = 32 samples x 64 output features; the input is fp16 (a conv / pool / mask
// output) or fp32 (a previous dense layer).
typedef _Float16 half8_d __attribute__((ext_vector_type(8)));

template <typename TIN, bool BF>
__global__ __launch_bounds__(64) void dense_f32_kernel(const TIN* __restrict__ in, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out, int N,
                                                       int in_mod, int K, int Cout, int relu, SiteArgs site, int B, int t0) {
    constexpr int RT = 2;                       // 64 output features per wave: two waves per SIMD at N = 7500, Cout = 512
    const int lane = threadIdx.x;
    const int r = lane & 31, hh = lane >> 5;
    const int n = blockIdx.x * 32 + r;
    const bool valid = n < N;
    const int c0 = blockIdx.y * (32 * RT);
    const int kh = K >> 1;
    const TIN* ip = in + (size_t)(valid ? n % in_mod : 0) * K + hh * kh;
    const float* wp = w + (size_t)(c0 + r) * K + hh * kh;

    f32x16 acc[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    // operands of step s+8 are requested before the 16 MFMAs (1024 cycles) of step s
    float b[8], a[RT][8];
#define DENSE_LOAD(S)                                                                                                  \
    {                                                                                                                  \
        if constexpr (sizeof(TIN) == 2) {                                                                              \
            const half8_d h_ = *(const half8_d*)(ip + (S));                                                            \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) b[e] = a16_to_f32<BF>(h_[e]);                                \
        } else {                                                                                                       \
            const float4 b0_ = *(const float4*)(ip + (S)), b1_ = *(const float4*)(ip + (S) + 4);                       \
            b[0] = b0_.x; b[1] = b0_.y; b[2] = b0_.z; b[3] = b0_.w; b[4] = b1_.x; b[5] = b1_.y; b[6] = b1_.z; b[7] = b1_.w; \
        }                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < RT; ++i) {                                                               \
            const float4 a0_ = *(const float4*)(wp + (size_t)(32 * i) * K + (S)), a1_ = *(const float4*)(wp + (size_t)(32 * i) * K + (S) + 4); \
            a[i][0] = a0_.x; a[i][1] = a0_.y; a[i][2] = a0_.z; a[i][3] = a0_.w;                                        \
            a[i][4] = a1_.x; a[i][5] = a1_.y; a[i][6] = a1_.z; a[i][7] = a1_.w;                                        \
        }                                                                                                              \
    }
    DENSE_LOAD(0);
    for (int s = 0; s < kh; s += 8) {
        float bc[8], ac[RT][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            bc[e] = valid ? b[e] : 0.f;
#pragma unroll
            for (int i = 0; i < RT; ++i) ac[i][e] = a[i][e];
        }
        if (s + 8 < kh) DENSE_LOAD(s + 8);
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int i = 0; i < RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[i][e], bc[e], acc[i], 0, 0, 0);
    }
#undef DENSE_LOAD
    if (!valid) return;
    const int tl = n / B, bimg = n - tl * B;
    const uint32_t t = (uint32_t)(t0 + tl);
    const float* mrow = site.kind == BMI_SITE_MASKSEMBLE ? site.masks + (size_t)((site.cnt0 + (int)t) % site.num_masks) * Cout : nullptr;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = c0 + 32 * i + 8 * q + 4 * hh;      // 4 consecutive output features: registers 4q .. 4q+3
            const float4 bi = *(const float4*)(bias + c);
            float v[4] = {acc[i][4 * q] + bi.x, acc[i][4 * q + 1] + bi.y, acc[i][4 * q + 2] + bi.z, acc[i][4 * q + 3] + bi.w};
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (site.kind == BMI_SITE_ELEMENTWISE || site.kind == BMI_SITE_CHANNEL) {
                // [B, Cout] tensor: element = b * Cout + c (a per-(image, channel) draw is the same thing here)
                const uint64_t elem = (uint64_t)bimg * Cout + c;
                const uint32_t keep = site_keep8(site, elem & ~(uint64_t)7, t) >> (elem & 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ((keep >> e) & 1u) ? v[e] * site.scale : 0.f;
            } else if (site.kind == BMI_SITE_MASKSEMBLE) {
                const float4 m4 = *(const float4*)(mrow + c);
                v[0] *= m4.x; v[1] *= m4.y; v[2] *= m4.z; v[3] *= m4.w;
            }
            *(float4*)(out + (size_t)n * Cout + c) = make_float4(v[0], v[1], v[2], v[3]);
        }
}

int launch_dense_f32(const void* in, int in_kind, const float* w, const float* bias, float* out, int n, int in_mod, int k,
                     int cout, int relu, const SiteArgs& site, int batch, int t0, hipStream_t s) {
    if (n <= 0 || in_mod <= 0 || batch <= 0) return BMI_ERR_INVALID;
    if (k % 16 != 0 || cout % 64 != 0) return BMI_ERR_UNSUPPORTED;
    const dim3 grid((n + 31) / 32, cout / 64), block(64);
    if (in_kind == 1) hipLaunchKernelGGL((dense_f32_kernel<float, false>), grid, block, 0, s, (const float*)in, w, bias, out, n, in_mod, k, cout, relu, site, batch, t0);
    else if (in_kind == 2) hipLaunchKernelGGL((dense_f32_kernel<_Float16, true>), grid, block, 0, s, (const _Float16*)in, w, bias, out, n, in_mod, k, cout, relu, site, batch, t0);
    else if (in_kind == 0) hipLaunchKernelGGL((dense_f32_kernel<_Float16, false>), grid, block, 0, s, (const _Float16*)in, w, bias, out, n, in_mod, k, cout, relu, site, batch, t0);
    else return BMI_ERR_INVALID;
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}
