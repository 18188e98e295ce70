// Exit classifier: logits = feat . W^T + bias, probs = softmax(logits)      (fp32 throughout)
// (ex{1,2,3}linear / linear, SA/models/resnet18/resnet18.py:314,:325,:335,:344; softmax of
//  FullAnalysis._get_output, SA/train/results_analyzer.py:242).
//
// One wave per 32 image-samples.  v_mfma_f32_32x32x2_f32 (exact f32 FMA chain) with CLASSES on
// the MFMA row axis and SAMPLES on the column axis: a lane owns one sample and 16 classes per
// 32-class tile, so the softmax max/sum are in-lane reductions plus ONE cross-half shuffle
// (lane ^ 32).  The K index is permuted (lane half h takes k in [h*K/2, (h+1)*K/2)), which a
// dot product does not care about, so every lane streams its row with float4 loads.
#include "kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int RT>
__global__ __launch_bounds__(64) void linear_softmax_kernel(const float* __restrict__ feat, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ logits,
                                                            float* __restrict__ probs, int N, int K, int C, SiteArgs site,
                                                            int B, int t0) {
    const int lane = threadIdx.x;
    const int r = lane & 31, hh = lane >> 5;
    const int n = blockIdx.x * 32 + r;
    const bool valid = n < N;
    const int kh = K >> 1;
    const float* fp = feat + (size_t)(valid ? n : 0) * K + hh * kh;
    const float* wp = w + (size_t)r * K + hh * kh;

    f32x16 acc[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    for (int s = 0; s < kh; s += 4) {
        float4 b = *(const float4*)(fp + s);
        if (!valid) b = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const float4 a = *(const float4*)(wp + (size_t)(32 * i) * K + s);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[i], 0, 0, 0);
        }
    }

    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * hh;
            if (c < C) {
                float v = acc[i][e] + bias[c];
                if (site.kind == BMI_SITE_ELEMENTWISE) {
                    // dropout on the logits (converter/pytorch wraps the last Linear too, nn2bnn.py:33-45):
                    // [B, C] tensor, element = b*C + c, eight elements per Philox call
                    const int tl = n / B;
                    const uint64_t elem = (uint64_t)(n - tl * B) * C + c;
                    const uint32_t keep = site_keep8(site, elem & ~(uint64_t)7, (uint32_t)(t0 + tl));
                    v = ((keep >> (elem & 7)) & 1u) ? v * site.scale : 0.f;
                }
                acc[i][e] = v;
                mx = fmaxf(mx, v);
                if (valid) logits[(size_t)n * C + c] = v;
            }
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * hh;
            if (c < C) {
                const float ex = expf(acc[i][e] - mx);
                acc[i][e] = ex;
                sum += ex;
            }
        }
    sum += __shfl_xor(sum, 32);
    if (!valid) return;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * hh;
            if (c < C) probs[(size_t)n * C + c] = acc[i][e] / sum;
        }
}

int launch_linear_softmax(const float* feat, const float* w, const float* bias, float* logits, float* probs, int n,
                          int k, int out_dim, const SiteArgs& site, int batch, int t0, hipStream_t s) {
    if (n <= 0 || out_dim <= 0 || batch <= 0) return BMI_ERR_INVALID;
    if (site.kind != BMI_SITE_NONE && site.kind != BMI_SITE_ELEMENTWISE) return BMI_ERR_UNSUPPORTED;
    if (k % 8 != 0 || out_dim > 128) return BMI_ERR_UNSUPPORTED;
    const int rt = (out_dim + 31) / 32;
    const dim3 grid((n + 31) / 32), block(64);
    switch (rt) {
        case 1: hipLaunchKernelGGL(linear_softmax_kernel<1>, grid, block, 0, s, feat, w, bias, logits, probs, n, k, out_dim, site, batch, t0); break;
        case 2: hipLaunchKernelGGL(linear_softmax_kernel<2>, grid, block, 0, s, feat, w, bias, logits, probs, n, k, out_dim, site, batch, t0); break;
        case 3: hipLaunchKernelGGL(linear_softmax_kernel<3>, grid, block, 0, s, feat, w, bias, logits, probs, n, k, out_dim, site, batch, t0); break;
        default: hipLaunchKernelGGL(linear_softmax_kernel<4>, grid, block, 0, s, feat, w, bias, logits, probs, n, k, out_dim, site, batch, t0); break;
    }
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}
