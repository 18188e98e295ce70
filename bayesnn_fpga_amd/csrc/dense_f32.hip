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

// Tiling (round 2): one workgroup = 64 output features x 128 samples, 4 waves (each 64 x 32: two 32x32 accumulators), K in
// chunks of 32 staged through LDS.  The first version gave every lane its own weight row and its own input row (float4 loads
// at a 2 KB stride: 64 cache lines per wave instruction, ~3000 line accesses per 1024 MFMA cycles and CU): 69-80 us per
// 7500 x 512 x 512 layer against 31 us of exact-f32 MFMA time.  Here a chunk is fetched with 8 lanes per 128-byte row segment
// and transposed by LDS: rows are stored [even k | odd k] (+4 floats of padding: 16 consecutive lanes hit 16 different 16-byte
// slots), so lane half h reads its sixteen k = 2j + h of the chunk as four ds_read_b128.
#define DN_KC 32
#define DN_ROW (DN_KC + 4)
#define DN_TF 64
#define DN_TS 128

template <typename TIN, bool BF>
__global__ __launch_bounds__(256) void dense_f32_kernel(const TIN* __restrict__ in, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out, int N,
                                                        int in_mod, int K, int Cout, int relu, SiteArgs site, int B, int t0) {
    constexpr int RT = DN_TF / 32;
    __shared__ __attribute__((aligned(16))) float Wt[DN_TF * DN_ROW];
    __shared__ __attribute__((aligned(16))) float It[DN_TS * DN_ROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int n0 = blockIdx.x * DN_TS, c0 = blockIdx.y * DN_TF;
    const int n = n0 + wave * 32 + r;
    const bool valid = n < N;

    f32x16 acc[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    // staging: weights 64 rows x 8 float4 = 512 pieces (2 per thread); input 128 rows x 32 k: fp32 1024 float4 (4 per thread),
    // 16-bit 512 pieces of 8 (2 per thread); rows beyond N read row 0 (computed, never stored)
    typedef float f32x4_d __attribute__((ext_vector_type(4)));
    typedef float f32x2_d __attribute__((ext_vector_type(2)));
    // the next chunk is fetched into registers while the MFMAs of the current one run
    f32x4_d wv[2];
    half8_d xh[2];
    f32x4_d xf[4];
#define DN_FETCH(K0)                                                                                               \
    {                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                            \
            const int f = tid + 256 * i, row = f >> 3, kq = f & 7;                                                 \
            wv[i] = *(const f32x4_d*)(w + (size_t)(c0 + row) * K + (K0) + 4 * kq);                                 \
        }                                                                                                          \
        if constexpr (sizeof(TIN) == 2) {                                                                          \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                        \
                const int f = tid + 256 * i, row = f >> 2, k8 = f & 3;                                             \
                const int nn = n0 + row < N ? n0 + row : 0;                                                        \
                xh[i] = *(const half8_d*)(in + (size_t)(nn % in_mod) * K + (K0) + 8 * k8);                         \
            }                                                                                                      \
        } else {                                                                                                   \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                        \
                const int f = tid + 256 * i, row = f >> 3, kq = f & 7;                                             \
                const int nn = n0 + row < N ? n0 + row : 0;                                                        \
                xf[i] = *(const f32x4_d*)((const float*)in + (size_t)(nn % in_mod) * K + (K0) + 4 * kq);           \
            }                                                                                                      \
        }                                                                                                          \
    }
    DN_FETCH(0);
    for (int k0 = 0; k0 < K; k0 += DN_KC) {
        __syncthreads();                 // the previous chunk's fragment reads are done
        if constexpr (sizeof(TIN) == 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int f = tid + 256 * i, row = f >> 2, k8 = f & 3;
                float* dst = It + row * DN_ROW + 4 * k8;
                *(f32x4_d*)dst = f32x4_d{a16_to_f32<BF>(xh[i][0]), a16_to_f32<BF>(xh[i][2]), a16_to_f32<BF>(xh[i][4]), a16_to_f32<BF>(xh[i][6])};
                *(f32x4_d*)(dst + 16) = f32x4_d{a16_to_f32<BF>(xh[i][1]), a16_to_f32<BF>(xh[i][3]), a16_to_f32<BF>(xh[i][5]), a16_to_f32<BF>(xh[i][7])};
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = tid + 256 * i, row = f >> 3, kq = f & 7;
                float* dst = It + row * DN_ROW + 2 * kq;
                *(f32x2_d*)dst = f32x2_d{xf[i][0], xf[i][2]};
                *(f32x2_d*)(dst + 16) = f32x2_d{xf[i][1], xf[i][3]};
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i, row = f >> 3, kq = f & 7;
            float* dst = Wt + row * DN_ROW + 2 * kq;
            *(f32x2_d*)dst = f32x2_d{wv[i][0], wv[i][2]};
            *(f32x2_d*)(dst + 16) = f32x2_d{wv[i][1], wv[i][3]};
        }
        __syncthreads();
        if (k0 + DN_KC < K) DN_FETCH(k0 + DN_KC);
        f32x4_d bq[4], aq[RT][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bq[q] = *(const f32x4_d*)(It + (wave * 32 + r) * DN_ROW + 16 * hh + 4 * q);
#pragma unroll
            for (int i = 0; i < RT; ++i) aq[i][q] = *(const f32x4_d*)(Wt + (32 * i + r) * DN_ROW + 16 * hh + 4 * q);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[i][q][e], bq[q][e], acc[i], 0, 0, 0);
    }
#undef DN_FETCH
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

// ---------------------------------------------------------------------------------------------------------------
// Split-fp16 form (round 2, the default for fp16 / fp32 inputs): the exact-f32 MFMA runs at 1/16 of the fp16 rate and a
// 7500 x 512 x 512 layer cannot take less than 29 us on it (53 measured).  Here every fp32 operand is split into an fp16 head
// and an fp16 tail (v = hi + lo, lo = fp16(v - hi): 22 bits of mantissa) while it is staged into LDS, and the product is
// hi*hi + lo*hi + hi*lo on v_mfma_f32_16x16x32_f16 with fp32 accumulation (an fp16 input IS its head: two MFMAs): each fp16
// product is exact in fp32, what is dropped is lo*lo (2^-22 relative) — fp32-equivalent to a few 1e-7, 5-8x the MFMA rate.
// Not for operands beyond the fp16 range (|v| > 65504): dense activations / weights of the path are O(1).  The exact kernel
// above stays for bf16 inputs and behind bmi_set_option("dense_exact", 1).
typedef _Float16 half4_d __attribute__((ext_vector_type(4)));

// XKIND: 0 fp16 input (its own head), 1 fp32, 3 | 4 a pair32 tensor of the split engines (fp16 | bf16 halves: decoded while it is fetched)
template <int XKIND>
__global__ __launch_bounds__(256) void dense_split_kernel(const void* __restrict__ in_, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out, int N,
                                                          int in_mod, int K, int Cout, int relu, SiteArgs site, int B, int t0) {
    constexpr bool XF32 = XKIND != 0;
    __shared__ __attribute__((aligned(16))) _Float16 Wh[DN_TF * 32], Wl[DN_TF * 32], Xh[DN_TS * 32], Xl[XF32 ? DN_TS * 32 : 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * DN_TS, c0 = blockIdx.y * DN_TF;
    typedef float f32x4_s __attribute__((ext_vector_type(4)));
    f32x4_s acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_s{0.f, 0.f, 0.f, 0.f};
    f32x4_s wv[2], xf[XF32 ? 4 : 1];
    half8_d xh[XF32 ? 1 : 2];
#define DS_FETCH(K0)                                                                                               \
    {                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                            \
            const int f = tid + 256 * i, row = f >> 3, q4 = f & 7;                                                 \
            wv[i] = *(const f32x4_s*)(w + (size_t)(c0 + row) * K + (K0) + 4 * q4);                                 \
        }                                                                                                          \
        if constexpr (XF32) {                                                                                      \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                        \
                const int f = tid + 256 * i, row = f >> 3, q4 = f & 7;                                             \
                const int nn = n0 + row < N ? n0 + row : 0;                                                        \
                if constexpr (XKIND >= 3) {                                                                        \
                    float d_[4];                                                                                   \
                    pair_decode<XKIND == 4, 4>((const _Float16*)in_ + pair32_off((size_t)(nn % in_mod), K, (K0) + 4 * q4), d_); \
                    xf[i] = f32x4_s{d_[0], d_[1], d_[2], d_[3]};                                                   \
                } else                                                                                             \
                xf[i] = *(const f32x4_s*)((const float*)in_ + (size_t)(nn % in_mod) * K + (K0) + 4 * q4);          \
            }                                                                                                      \
        } else {                                                                                                   \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                        \
                const int f = tid + 256 * i, row = f >> 2, k8 = f & 3;                                             \
                const int nn = n0 + row < N ? n0 + row : 0;                                                        \
                xh[i] = *(const half8_d*)((const _Float16*)in_ + (size_t)(nn % in_mod) * K + (K0) + 8 * k8);       \
            }                                                                                                      \
        }                                                                                                          \
    }
#define DS_SPLIT(V, HI, LO)                                                                                        \
    {                                                                                                              \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                            \
            HI[e] = (_Float16)(V)[e];                                                                              \
            LO[e] = (_Float16)((V)[e] - (float)HI[e]);                                                             \
        }                                                                                                          \
    }
    DS_FETCH(0);
    for (int k0 = 0; k0 < K; k0 += 32) {
        __syncthreads();                 // the previous chunk's fragment reads are done
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i, row = f >> 3, q4 = f & 7;
            half4_d hi, lo;
            DS_SPLIT(wv[i], hi, lo);
            *(half4_d*)(Wh + row * 32 + 4 * q4) = hi;
            *(half4_d*)(Wl + row * 32 + 4 * q4) = lo;
        }
        if constexpr (XF32) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = tid + 256 * i, row = f >> 3, q4 = f & 7;
                half4_d hi, lo;
                DS_SPLIT(xf[i], hi, lo);
                *(half4_d*)(Xh + row * 32 + 4 * q4) = hi;
                *(half4_d*)(Xl + row * 32 + 4 * q4) = lo;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int f = tid + 256 * i, row = f >> 2, k8 = f & 3;
                *(half8_d*)(Xh + row * 32 + 8 * k8) = xh[i];
            }
        }
        __syncthreads();
        if (k0 + 32 < K) DS_FETCH(k0 + 32);
        half8_t ah[4], al[4], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ah[i] = *(const half8_t*)(Wh + (16 * i + l16) * 32 + 8 * kq);
            al[i] = *(const half8_t*)(Wl + (16 * i + l16) * 32 + 8 * kq);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bh[j] = *(const half8_t*)(Xh + (wave * 32 + 16 * j + l16) * 32 + 8 * kq);
            if constexpr (XF32) bl[j] = *(const half8_t*)(Xl + (wave * 32 + 16 * j + l16) * 32 + 8 * kq);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = mfma_16x16x32<false>(ah[i], bh[j], acc[i][j]);
                acc[i][j] = mfma_16x16x32<false>(al[i], bh[j], acc[i][j]);
                if constexpr (XF32) acc[i][j] = mfma_16x16x32<false>(ah[i], bl[j], acc[i][j]);
            }
    }
#undef DS_FETCH
#undef DS_SPLIT
    // lane = sample l16 of tile j, registers = 4 consecutive features 16*i + 4*kq ..
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wave * 32 + 16 * j + l16;
        if (n >= N) continue;
        const int tl = n / B, bimg = n - tl * B;
        const uint32_t t = (uint32_t)(t0 + tl);
        const float* mrow = site.kind == BMI_SITE_MASKSEMBLE ? site.masks + (size_t)((site.cnt0 + (int)t) % site.num_masks) * Cout : nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + 16 * i + 4 * kq;
            const float4 bi = *(const float4*)(bias + c);
            float v[4] = {acc[i][j][0] + bi.x, acc[i][j][1] + bi.y, acc[i][j][2] + bi.z, acc[i][j][3] + bi.w};
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (site.kind == BMI_SITE_ELEMENTWISE || site.kind == BMI_SITE_CHANNEL) {
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
}

int launch_dense_f32(const void* in, int in_kind, const float* w, const float* bias, float* out, int n, int in_mod, int k,
                     int cout, int relu, const SiteArgs& site, int batch, int t0, hipStream_t s) {
    if (n <= 0 || in_mod <= 0 || batch <= 0) return BMI_ERR_INVALID;
    if (k % DN_KC != 0 || cout % DN_TF != 0) return BMI_ERR_UNSUPPORTED;
    const dim3 grid((n + DN_TS - 1) / DN_TS, cout / DN_TF), block(256);
    if ((!opt_dense_exact() && in_kind != 2) || in_kind >= 3) {     // (pair32 inputs: the split form always)
        if (in_kind == 1) hipLaunchKernelGGL((dense_split_kernel<1>), grid, block, 0, s, in, w, bias, out, n, in_mod, k, cout, relu, site, batch, t0);
        else if (in_kind == 0) hipLaunchKernelGGL((dense_split_kernel<0>), grid, block, 0, s, in, w, bias, out, n, in_mod, k, cout, relu, site, batch, t0);
        else if (in_kind == 3) hipLaunchKernelGGL((dense_split_kernel<3>), grid, block, 0, s, in, w, bias, out, n, in_mod, k, cout, relu, site, batch, t0);
        else if (in_kind == 4) hipLaunchKernelGGL((dense_split_kernel<4>), grid, block, 0, s, in, w, bias, out, n, in_mod, k, cout, relu, site, batch, t0);
        else return BMI_ERR_INVALID;
        BMI_CHECK_LAUNCH();
        return BMI_OK;
    }
    if (in_kind == 1) hipLaunchKernelGGL((dense_f32_kernel<float, false>), grid, block, 0, s, (const float*)in, w, bias, out, n, in_mod, k, cout, relu, site, batch, t0);
    else if (in_kind == 2) hipLaunchKernelGGL((dense_f32_kernel<_Float16, true>), grid, block, 0, s, (const _Float16*)in, w, bias, out, n, in_mod, k, cout, relu, site, batch, t0);
    else if (in_kind == 0) hipLaunchKernelGGL((dense_f32_kernel<_Float16, false>), grid, block, 0, s, (const _Float16*)in, w, bias, out, n, in_mod, k, cout, relu, site, batch, t0);
    else return BMI_ERR_INVALID;
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}
