// The EXACT engine's kernels (bmi_model_desc.dtype = BMI_DTYPE_F32): fp32 activations, fp32 weights, every product an fp32
// product accumulated in fp32 on v_mfma_f32_32x32x2_f32.  Not a speed path (1/16 of the fp16 MFMA rate, register-staged single
// buffer): it is the arithmetic of the reference's own CPU path (fp32 ATen conv2d / batch_norm / linear, SA/models/resnet18/
// resnet18.py:32-48, :302-346) on the device, so that a parity test can tell rounding from a bug — SURVEY.md §7 hard part 2
// ("keep an fp32-MFMA path for parity tests").  One generic per-tap implicit-GEMM conv with the whole fused epilogue
// (folded BN, inner / outer stochastic site of every kind, residual, ReLU), plus the fp32 forms of the stem, the stand-alone
// site and the 2x2 max-pool.  Dense layers and heads are fp32 already (dense_f32.hip, head_fused.hip).
//
//   D[cout][pixel] = sum_k W[cout][k] * X[k][pixel],   k = (ky*ks + kx)*Cin + ci      (as conv_igemm.hip)
//
// Tile 64 channels x 64 pixels x 32 deep, 256 threads: wave w = channel half w >> 1, pixel half w & 1, one 32 x 32
// accumulator.  A lane holds pixel (lane & 31) and, per accumulator quad q, the four CONSECUTIVE channels 8 q + 4 (lane >> 5) ..:
// the per-quad epilogue of conv_epilogue.h (site_mult4) applies unchanged.  Within a 32-deep K-step lane half h supplies
// k = 8 m + 4 h + e to the e-th MFMA of group m for BOTH operands (one ds_read_b128 each per four MFMAs): a permutation of the
// K order, which a sum does not see.
#include "conv_epilogue.h"
#include "kernels.h"

typedef float f32x16_x __attribute__((ext_vector_type(16)));
typedef float f32x4_x __attribute__((ext_vector_type(4)));

#define XK 32            // K-step depth (channels of one tap)
#define XROW (XK + 4)    // LDS row pitch in floats: 144 B, rows 16-byte aligned, consecutive rows 4 banks apart

__global__ __launch_bounds__(256) void conv_exact_kernel(ConvArgs a) {
    __shared__ __attribute__((aligned(16))) float As[64 * XROW];
    __shared__ __attribute__((aligned(16))) float Bs[64 * XROW];
    const float* const in = (const float*)a.in;
    const float* const wgt = (const float*)a.wgt;
    const float* const res = (const float*)a.res;
    float* const out = (float*)a.out;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int wc = wave >> 1, wp = wave & 1;
    const int ch0 = blockIdx.y * 64;
    const long pix0 = (long)blockIdx.x * 64;
    const int HoWo = a.Ho * a.Wo;
    const int Ktot = a.ksize * a.ksize * a.Cin;

    // staging: thread -> row (tid >> 2), 8 consecutive k (two float4) at 8 * (tid & 3)
    const int srow = tid >> 2, sk = (tid & 3) * 8;
    const float* wrow = wgt + (size_t)(ch0 + srow) * Ktot + sk;
    const long m_s = pix0 + srow;
    const bool vm = m_s < a.M;
    int iy0 = 0, ix0 = 0;
    const float* xbase = in;
    if (vm) {
        const int n = (int)(m_s / HoWo), rem = (int)(m_s - (long)n * HoWo);
        const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
        iy0 = oy * a.stride - a.pad;
        ix0 = ox * a.stride - a.pad;
        xbase = in + (size_t)(n % a.in_mod) * a.H * a.W * a.Cin + sk;
    }

    f32x16_x acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;

    for (int tap = 0; tap < a.ksize * a.ksize; ++tap) {
        const int ky = tap / a.ksize, kx = tap - ky * a.ksize;
        const int iy = iy0 + ky, ix = ix0 + kx;
        const bool ok = vm && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const float* xp = xbase + ((size_t)iy * a.W + ix) * a.Cin;
        for (int c0 = 0; c0 < a.Cin; c0 += XK) {
            const f32x4_x w0 = *(const f32x4_x*)(wrow + tap * a.Cin + c0), w1 = *(const f32x4_x*)(wrow + tap * a.Cin + c0 + 4);
            f32x4_x x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0;
            if (ok) { x0 = *(const f32x4_x*)(xp + c0); x1 = *(const f32x4_x*)(xp + c0 + 4); }
            __syncthreads();   // the previous step's fragment reads are done
            *(f32x4_x*)(As + srow * XROW + sk) = w0; *(f32x4_x*)(As + srow * XROW + sk + 4) = w1;
            *(f32x4_x*)(Bs + srow * XROW + sk) = x0; *(f32x4_x*)(Bs + srow * XROW + sk + 4) = x1;
            __syncthreads();
#pragma unroll
            for (int m = 0; m < XK / 8; ++m) {
                const f32x4_x af = *(const f32x4_x*)(As + (wc * 32 + r) * XROW + 8 * m + 4 * hh);
                const f32x4_x bf = *(const f32x4_x*)(Bs + (wp * 32 + r) * XROW + 8 * m + 4 * hh);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc, 0, 0, 0);
            }
        }
    }

    // ---- epilogue: this lane's pixel, four channel quads (arithmetic and order of epilogue_quad, fp32 in and out) ----
    const long m_o = pix0 + wp * 32 + r;
    if (m_o >= a.M) return;
    const int n = (int)(m_o / HoWo), rem = (int)(m_o - (long)n * HoWo);
    PixelCtx p;
    p.out_off = ((size_t)n * HoWo + rem) * a.Cout;
    p.resp = nullptr;
    const int tl = n / a.B;
    p.b = n - tl * a.B;
    p.t = a.t0 + tl;
    p.e_pix = p.b * HoWo + rem;
    p.mrow = a.site.kind == BMI_SITE_MASKSEMBLE ? a.site.masks + (size_t)((a.site.cnt0 + p.t) % a.site.num_masks) * a.Cout : nullptr;
    const float* resp = res ? res + ((size_t)(n % a.res_mod) * HoWo + rem) * a.Cout : nullptr;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c4 = ch0 + wc * 32 + 8 * q + 4 * hh;
        float v[4] = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
        epilogue_quad_f32(a, p, resp, v, c4);
        *(f32x4_x*)(out + p.out_off + c4) = f32x4_x{v[0], v[1], v[2], v[3]};
    }
}

int launch_conv_exact(const ConvArgs& a, hipStream_t s) {
    if (a.Cin % XK != 0 || a.Cout % 64 != 0) return BMI_ERR_UNSUPPORTED;
    // the fp16 engine's launch forms the exact engine never builds (bmi_create keeps them out): fused shortcut, pair, input-side
    // keep bits, fused pooling, split-K, dynamic-exit row tables
    if (a.in2 || a.wgt_b || a.in_bits || a.in2_bits || a.pool || a.pool_b || a.partial || a.imap) return BMI_ERR_UNSUPPORTED;
    if (a.N <= 0 || a.M <= 0 || a.in_mod <= 0 || a.B <= 0 || (a.res && a.res_mod <= 0)) return BMI_ERR_INVALID;
    const dim3 grid((unsigned)((a.M + 63) / 64), (unsigned)(a.Cout / 64)), block(256);
    hipLaunchKernelGGL(conv_exact_kernel, grid, block, 0, s, a);
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}

// ---------------------------------------------------------------------------------------------
// stand-alone site on an fp32 tensor (mask_apply_kernel's arithmetic: MCDropout / Masksembles2D on a stage output, or the
// mask + BN shift + ReLU half of a deterministic conv with an inner site)
// PAIR = 0: fp32 tensors; 1 | 2: pair32 tensors (conv_epilogue.h) of fp16 | bf16 halves — the split engines
template <int PAIR>
__global__ __launch_bounds__(256) void mask_apply_f32_kernel(EltArgs a) {
    const float* const in = (const float*)a.in;
    float* const out = (float*)a.out;
    const int cg = a.C >> 3;
    const long total = (long)a.N * a.HW * cg;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cg) * 8;
        const long pix = i / cg;
        const int n = (int)(pix / a.HW);
        const int p = (int)(pix - (long)n * a.HW);
        const int tl = n / a.B, b = n - tl * a.B;
        const int t = a.t0 + tl;
        float v[8];
        if constexpr (PAIR == 0) {
            const float* src = in + ((size_t)(n % a.in_mod) * a.HW + p) * a.C + c8;
            const f32x4_x x0 = *(const f32x4_x*)src, x1 = *(const f32x4_x*)(src + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = x0[e]; v[4 + e] = x1[e]; }
        } else {
            pair_decode<PAIR == 2, 8>((const _Float16*)a.in + pair32_off((size_t)(n % a.in_mod) * a.HW + p, a.C, c8), v);
        }
        const uint64_t elem0 = a.site.kind == BMI_SITE_CHANNEL ? (uint64_t)b * a.C + c8 : ((uint64_t)b * a.HW + p) * a.C + c8;
        if (a.site.kind == BMI_SITE_ELEMENTWISE || a.site.kind == BMI_SITE_CHANNEL) {
            const uint32_t keep = site_keep8(a.site, elem0, (uint32_t)t);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = ((keep >> e) & 1u) ? v[e] * a.site.scale : 0.f;
        } else if (a.site.kind == BMI_SITE_MASKSEMBLE) {
            const float* mrow = a.site.masks + (size_t)((a.site.cnt0 + t) % a.site.num_masks) * a.C + c8;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= mrow[e];
        }
        if (a.bias_post) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += a.bias_post[c8 + e];
        }
        if (a.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if constexpr (PAIR == 0) {
            float* dst = out + ((size_t)n * a.HW + p) * a.C + c8;
            *(f32x4_x*)dst = f32x4_x{v[0], v[1], v[2], v[3]};
            *(f32x4_x*)(dst + 4) = f32x4_x{v[4], v[5], v[6], v[7]};
        } else {
            pair_encode<PAIR == 2, 8>((_Float16*)a.out + pair32_off((size_t)n * a.HW + p, a.C, c8), v);
        }
    }
}

// The first "block" site of the split engines (an elementwise 2-bit site on a deterministic tensor: B images in, chunk x B images out, 6.5 GB of
// pair32 on the headline) with ONE Philox call per 64 elements instead of one per 8: a wave owns 512 consecutive 8-channel items (= 64 calls, one per
// lane), and in pass k lane l takes item 64 k + l, whose call lane 8 k + (l >> 3) holds (four ds_bpermute).  The generic kernel above spends more
// than half of its VALU time recomputing each call eight times.  Launcher: a sample's items are a multiple of 512 (a wave never straddles two
// samples), the site's index offset a whole number of calls.
template <bool BF>
__global__ __launch_bounds__(256) void mask_apply_pair_lb1_kernel(EltArgs a) {
    const int cg = a.C >> 3;
    const long per_sample = (long)a.B * a.HW * cg;
    const long waves = per_sample / 512;                             // input tiles of 512 items
    const int T = a.N / a.B;
    const int lane = threadIdx.x & 63;
    const int t_lo = blockIdx.y * a.tchunk, t_hi = min(T, t_lo + a.tchunk);
    for (long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6); w < waves; w += (long)gridDim.x * 4) {
        const long r0 = w * 512;                                     // item index inside a sample
        // the tile's input, decoded once for all the samples of this workgroup's range (the deterministic tensor is B images: read per
        // sample it was as many fabric bytes again as the output)
        float x[8][8];
        size_t ooff[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const long r = r0 + 64 * k + lane;                        // this lane's item: image b, pixel p, channels c8 ..
            const int c8 = (int)(r % cg) * 8;
            const long bp = r / cg;                                   // b * HW + p
            pair_decode<BF, 8>((const _Float16*)a.in + pair32_off((size_t)bp, a.C, c8), x[k]);
            ooff[k] = pair32_off((size_t)bp, a.C, c8);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[k][e] *= a.site.scale;
        }
        for (int tl = t_lo; tl < t_hi; ++tl) {
            const philox4 mine = philox_site_call(a.site, (uint64_t)(r0 + 8 * lane) * 8, (uint32_t)(a.t0 + tl));      // call r0 / 8 + lane
            _Float16* const ot = (_Float16*)a.out + (size_t)tl * a.B * a.HW * (2 * (size_t)a.C);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int src = 8 * k + (lane >> 3);
                philox4 c;
#pragma unroll
                for (int wd = 0; wd < 4; ++wd) c.w[wd] = (uint32_t)__shfl((int)mine.w[wd], src, 64);
                const uint32_t keep = a.site.drop_all ? 0u : philox_keep8(c, (uint32_t)((r0 + 64 * k + lane) * 8), a.site.log2_bits, a.site.thresh);
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = ((keep >> e) & 1u) ? x[k][e] : 0.f;
                pair_encode<BF, 8>(ot + ooff[k], v);
            }
        }
    }
}

int launch_mask_apply_f32(const EltArgs& a, hipStream_t s) {
    if (a.C % 8 != 0 || (a.pair && a.C % 32 != 0)) return BMI_ERR_UNSUPPORTED;
    if (a.N <= 0 || a.in_mod <= 0 || a.B <= 0) return BMI_ERR_INVALID;
    if (a.pair && a.site.kind == BMI_SITE_ELEMENTWISE && a.site.log2_bits == 1 && !a.bias_post && !a.relu && a.N % a.B == 0 && a.in_mod == a.B &&
        ((long)a.B * a.HW * (a.C >> 3)) % 512 == 0 && a.site.elem_off % 64 == 0) {
        const long waves = ((long)a.B * a.HW * (a.C >> 3)) / 512;
        const long blocks = (waves + 3) / 4;
        const int T = a.N / a.B;
        EltArgs b = a;
        // samples per workgroup: as many as still leave ~16 workgroups per CU (the input tile is read once per workgroup)
        b.tchunk = (int)std::max<long>(1, std::min<long>(T, (long)T * blocks / (256 * 16)));
        const dim3 grid((unsigned)std::min<long>(blocks, 0x7fffffffL), (unsigned)((T + b.tchunk - 1) / b.tchunk));
        if (a.pair == 1) hipLaunchKernelGGL(mask_apply_pair_lb1_kernel<false>, grid, dim3(256), 0, s, b);
        else hipLaunchKernelGGL(mask_apply_pair_lb1_kernel<true>, grid, dim3(256), 0, s, b);
        BMI_CHECK_LAUNCH();
        return BMI_OK;
    }
    long blocks = ((long)a.N * a.HW * (a.C >> 3) + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (a.pair == 1) hipLaunchKernelGGL(mask_apply_f32_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    else if (a.pair == 2) hipLaunchKernelGGL(mask_apply_f32_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(mask_apply_f32_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}

__global__ __launch_bounds__(256) void maxpool2_f32_kernel(const float* in, float* out, int N, int H, int W, int C) {
    const int cg = C >> 2, Ho = H >> 1, Wo = W >> 1;
    const long total = (long)N * Ho * Wo * cg;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % cg) * 4;
        long q = i / cg;
        const int ox = (int)(q % Wo); q /= Wo;
        const int oy = (int)(q % Ho);
        const int n = (int)(q / Ho);
        const float* p = in + (((size_t)n * H + 2 * oy) * W + 2 * ox) * C + c4;
        const f32x4_x a0 = *(const f32x4_x*)p, a1 = *(const f32x4_x*)(p + C);
        const f32x4_x a2 = *(const f32x4_x*)(p + (size_t)W * C), a3 = *(const f32x4_x*)(p + (size_t)W * C + C);
        f32x4_x o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf(fmaxf(a0[e], a1[e]), fmaxf(a2[e], a3[e]));
        *(f32x4_x*)(out + (((size_t)n * Ho + oy) * Wo + ox) * C + c4) = o;
    }
}

// pair32 tensors: the maximum of the four decoded values, re-encoded (it IS one of the four pairs' value, so the encoding is exact again)
template <bool BF>
__global__ __launch_bounds__(256) void maxpool2_pair_kernel(const _Float16* in, _Float16* out, int N, int H, int W, int C) {
    const int cg = C >> 3, Ho = H >> 1, Wo = W >> 1;
    const long total = (long)N * Ho * Wo * cg;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cg) * 8;
        long q = i / cg;
        const int ox = (int)(q % Wo); q /= Wo;
        const int oy = (int)(q % Ho);
        const int n = (int)(q / Ho);
        const size_t p00 = ((size_t)n * H + 2 * oy) * W + 2 * ox;
        float a0[8], a1[8], a2[8], a3[8], o[8];
        pair_decode<BF, 8>(in + pair32_off(p00, C, c8), a0);
        pair_decode<BF, 8>(in + pair32_off(p00 + 1, C, c8), a1);
        pair_decode<BF, 8>(in + pair32_off(p00 + W, C, c8), a2);
        pair_decode<BF, 8>(in + pair32_off(p00 + W + 1, C, c8), a3);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaxf(fmaxf(a0[e], a1[e]), fmaxf(a2[e], a3[e]));
        pair_encode<BF, 8>(out + pair32_off(((size_t)n * Ho + oy) * Wo + ox, C, c8), o);
    }
}

int launch_maxpool2_f32(const float* in, float* out, int n, int h, int w, int c, hipStream_t s, int pair) {
    if (c % 4 != 0 || (h & 1) || (w & 1) || (pair && c % 32 != 0)) return BMI_ERR_UNSUPPORTED;
    const long total = (long)n * (h / 2) * (w / 2) * (c / (pair ? 8 : 4));
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks <= 0) return BMI_ERR_INVALID;
    if (pair == 1) hipLaunchKernelGGL(maxpool2_pair_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, (const _Float16*)in, (_Float16*)out, n, h, w, c);
    else if (pair == 2) hipLaunchKernelGGL(maxpool2_pair_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, (const _Float16*)in, (_Float16*)out, n, h, w, c);
    else hipLaunchKernelGGL(maxpool2_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, out, n, h, w, c);
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}
