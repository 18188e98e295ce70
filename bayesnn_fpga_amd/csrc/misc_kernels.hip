// Non-GEMM kernels of the path (all HBM/L2-bound, 16-byte accesses per lane):
//   stem_conv      fp32 NCHW network input -> conv(Cin<=4)+BN(+ReLU) -> fp16 NHWC
//                  (ResNet stem, SA/models/resnet18/resnet18.py:303: no ReLU after bn1)
//   mask_apply     stand-alone stochastic site; also expands a deterministic tensor [B,...] to
//                  the folded [samples*B,...] batch (MCDropout :207-210 / Masksembles2D utils.py:165-169)
//   maxpool2       VGG block-end MaxPool2d(2,2) (SA/models/vgg19/vgg19.py:127)
//   finalize       mean / variance / mean logit
//   philox_mask    keep bits, for unit tests
#include <cstdlib>

#include "conv_epilogue.h"
#include "kernels.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void site_mask8(const SiteArgs& s, float v[8], uint64_t elem0, int t, const float* mrow) {
    // elem0: element index of v[0] (multiple of 8 within the site's index space)
    if (s.kind == BMI_SITE_ELEMENTWISE || s.kind == BMI_SITE_CHANNEL) {
        const uint32_t keep = site_keep8(s, elem0, (uint32_t)t);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ((keep >> e) & 1u) ? v[e] * s.scale : 0.f;
    } else if (s.kind == BMI_SITE_MASKSEMBLE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= mrow[e];
    }
}

// ---------------------------------------------------------------------------------------------
template <bool BF>
__global__ __launch_bounds__(256) void mask_apply_kernel(EltArgs a) {
    const int cg = a.C >> 3;  // 8-channel groups per pixel
    const long total = (long)a.N * a.HW * cg;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cg) * 8;
        const long pix = i / cg;  // n*HW + p
        const int n = (int)(pix / a.HW);
        const int p = (int)(pix - (long)n * a.HW);
        const int tl = n / a.B, b = n - tl * a.B;
        const int t = a.t0 + tl;
        const half8 x = *(const half8*)(a.in + ((size_t)(n % a.in_mod) * a.HW + p) * a.C + c8);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = a16_to_f32<BF>(x[e]);
        const uint64_t elem0 = a.site.kind == BMI_SITE_CHANNEL ? (uint64_t)b * a.C + c8
                                                                : ((uint64_t)b * a.HW + p) * a.C + c8;
        const float* mrow = a.site.kind == BMI_SITE_MASKSEMBLE
                                ? a.site.masks + (size_t)((a.site.cnt0 + t) % a.site.num_masks) * a.C + c8
                                : nullptr;
        site_mask8(a.site, v, elem0, t, mrow);
        if (a.bias_post) {   // inner site of a deterministic conv: the BN shift (and the ReLU) come after the mask
            const float4 p0 = *(const float4*)(a.bias_post + c8), p1 = *(const float4*)(a.bias_post + c8 + 4);
            v[0] += p0.x; v[1] += p0.y; v[2] += p0.z; v[3] += p0.w;
            v[4] += p1.x; v[5] += p1.y; v[6] += p1.z; v[7] += p1.w;
        }
        if (a.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = a16_from_f32<BF>(v[e]);
        *(half8*)((_Float16*)a.out + ((size_t)n * a.HW + p) * a.C + c8) = o;
    }
}

// Elementwise MC-dropout sites with fewer than 16 bits per element: one Philox call masks 128 / k elements, i.e. the
// items of R = 16 / k consecutive lanes.  A wave works on super-blocks of 64 calls (64 * R items, contiguous in the
// sample's NHWC element order): every lane runs Philox ONCE for call (super-block * 64 + lane), then for each of its
// R items fetches the four words of the owning lane with ds_bpermute.  Philox is 480 of the ~700 cycles a wave spends
// per item otherwise (v_mad_u64_u32 is quarter rate): at p = 0.25 (R = 8) this turns the kernel from Philox-bound into
// HBM-bound.  Needs the sample's element count to be a multiple of the super-block (64 * 128 / k elements).
template <int LB, bool BF>
__global__ __launch_bounds__(256) void mask_apply_shared_kernel(EltArgs a) {
    constexpr int R = 16 >> LB;                    // items (8 elements each) per call
    const int lane = threadIdx.x & 63;
    const long sample_elems = (long)a.B * a.HW * a.C;
    const long sb_per_sample = sample_elems / (64L * R * 8);
    const long n_sb = sb_per_sample * (a.N / a.B);
    const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((long)gridDim.x * blockDim.x) >> 6;
    const size_t in_sample_stride = a.in_mod == a.B ? 0 : (size_t)sample_elems;   // deterministic input: every sample reads [B]
    for (long sb = wave0; sb < n_sb; sb += n_waves) {
        const long tl = sb / sb_per_sample, sbs = sb - tl * sb_per_sample;
        const uint32_t t = (uint32_t)(a.t0 + tl);
        const uint64_t g = (uint64_t)sbs * 64 + lane + (a.site.elem_off >> (7 - LB));   // this lane's call
        const philox4 mine = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), t, (uint32_t)a.site.site_id, a.site.seed_lo,
                                           a.site.seed_hi);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int owner = j * (64 / R) + lane / R;
            philox4 r;
#pragma unroll
            for (int wd = 0; wd < 4; ++wd) r.w[wd] = (uint32_t)__shfl((int)mine.w[wd], owner, 64);
            const size_t e0 = ((size_t)sbs * 64 * R + (size_t)j * 64 + lane) * 8;    // element offset inside the sample
            const uint32_t keep = a.site.drop_all ? 0u : philox_keep8(r, (uint32_t)e0, LB, a.site.thresh);
            const half8 x = *(const half8*)(a.in + (size_t)tl * in_sample_stride + e0);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = ((keep >> e) & 1u) ? a16_to_f32<BF>(x[e]) * a.site.scale : 0.f;
            if (a.bias_post) {
                const int c8 = (int)(e0 % (size_t)a.C);
                const float4 p0 = *(const float4*)(a.bias_post + c8), p1 = *(const float4*)(a.bias_post + c8 + 4);
                v[0] += p0.x; v[1] += p0.y; v[2] += p0.z; v[3] += p0.w;
                v[4] += p1.x; v[5] += p1.y; v[6] += p1.z; v[7] += p1.w;
            }
            if (a.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = a16_from_f32<BF>(v[e]);
            *(half8*)((_Float16*)a.out + (size_t)tl * sample_elems + e0) = o;
        }
    }
}

// The MC-dropout case proper (2 bits per element: p = 0.25 / 0.5 / 0.75; no affine + ReLU behind the mask).  Two things kept
// mask_apply_shared_kernel at 4.2 TB/s of writes (0.78 ms for the 3.3 GB of the headline's first site; a fill reaches 6.9):
//   * vector ALU: four ds_bpermute per item to fetch the owner's whole Philox output, a compare + select per element.  Here every
//     lane parks its call's four words in a 1 KB per-wave LDS table (one ds_write_b128) and an item fetches just ITS 16 bits
//     (8 two-bit fields) with one ds_read_u16 (the LDS operations of one wave complete in order: no barrier); the keep test
//     runs on all 8 fields at once (with b0 / b1 the even / odd bits, field >= thresh is b1|b0, b1 or b1&b0 for thresh 1, 2,
//     3); a dropped element is cleared by ANDing the packed 16-bit results with a mask built from sign-extended bit
//     extracts.  0.75 -> 0.69 ms;
//   * fabric reads: see the work-item comment below.  0.69 -> 0.62 ms (ResNet-50's first site, 8.4 GB: 2.8 -> 1.6 ms).
// A kept element is x * scale in fp32 rounded once (the oracle's arithmetic).  (Letting the producing conv apply the scale so
// that this kernel is a pure AND was built and measured: no change, the conversions are not what it waits for.)
template <bool BF>
__global__ __launch_bounds__(256) void mask_apply_lb1_kernel(EltArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t words[4][64 * 4];
    const int lane = threadIdx.x & 63;
    uint32_t* const W = words[threadIdx.x >> 6];
    const long sample_elems = (long)a.B * a.HW * a.C;
    const long sb_per_sample = sample_elems / (64L * 8 * 8);
    const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((long)gridDim.x * blockDim.x) >> 6;
    const uint32_t thresh = a.site.thresh;
    const uint32_t use_or = thresh == 1 ? 0xFFFFu : 0u, use_and = thresh == 3 ? 0xFFFFu : 0u;
    const uint32_t keep_all = thresh == 0 ? 0x5555u : 0u;     // p = 0: every field >= 0 (the bit tricks below only cover thresh 1..3)
    const char* const myfield = (const char*)W + (lane >> 3) * 16 + 2 * (lane & 7);   // + 128 * j: owner lane j*8 + lane/8
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    // Work item = (super-block of the sample, chunk of `tchunk` samples).  With a deterministic input (the usual case: the site
    // expands the once-per-batch prefix to the folded batch) the item's 8 KB of input are loaded ONCE and masked tchunk times:
    // as one item per (sample, super-block) every sample re-read the whole [B] tensor through the fabric (33 MB does not fit an
    // XCD's 4 MB L2: 3.3 GB of Infinity-Cache reads next to the 3.3 GB of HBM writes, 4.8 TB/s of writes where a fill reaches
    // 6.9), and the kernel stayed at 0.68 ms however few vector instructions it issued.
    const int T = a.N / a.B;
    const int tchunk = a.in_mod == a.B ? a.tchunk : 1;
    const long n_tchunks = (T + tchunk - 1) / tchunk;
    const long n_items = sb_per_sample * n_tchunks;
    for (long it = wave0; it < n_items; it += n_waves) {
        const long tci = it / sb_per_sample, sbs = it - tci * sb_per_sample;
        const uint64_t g = (uint64_t)sbs * 64 + lane + (a.site.elem_off >> 6);   // this lane's call (2 bits per element: 64 per call)
        const int t_lo = (int)tci * tchunk, t_hi = min(T, t_lo + tchunk);
        half8 x[8];
        if (a.in_mod == a.B) {
            const _Float16* src = a.in + ((size_t)sbs * 512 + lane) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = *(const half8*)(src + j * 512);
        }
        for (int tl = t_lo; tl < t_hi; ++tl) {
            const philox4 mine = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)(a.t0 + tl), (uint32_t)a.site.site_id,
                                               a.site.seed_lo, a.site.seed_hi);
            *(u32x4*)(W + lane * 4) = u32x4{mine.w[0], mine.w[1], mine.w[2], mine.w[3]};
            if (a.in_mod != a.B) {
                const _Float16* src = a.in + (size_t)tl * sample_elems + ((size_t)sbs * 512 + lane) * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = *(const half8*)(src + j * 512);
            }
            _Float16* dst = (_Float16*)a.out + (size_t)tl * sample_elems + ((size_t)sbs * 512 + lane) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t f = *(const uint16_t*)(myfield + 128 * j);
                const uint32_t b0 = f & 0x5555u, b1 = (f >> 1) & 0x5555u;
                const uint32_t kb = a.site.drop_all ? 0u : (((b1 | (b0 & use_or)) & (b0 | ~use_and)) | keep_all);   // bit 2e = keep element e
                half8 r;
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] = a16_from_f32<BF>(a16_to_f32<BF>(x[j][e]) * a.site.scale);
                const u32x4 rb = __builtin_bit_cast(u32x4, r);
                u32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t lo = (uint32_t)(((int)(kb << (31 - 4 * i))) >> 31);        // element 2i   kept: all ones
                    const uint32_t hi = (uint32_t)(((int)(kb << (29 - 4 * i))) >> 31);        // element 2i+1
                    o[i] = rb[i] & ((lo & 0xFFFFu) | (hi & 0xFFFF0000u));
                }
                *(u32x4*)(dst + j * 512) = o;
            }
        }
    }
}

int launch_mask_apply(const EltArgs& a, hipStream_t s) {
    if (a.C % 8 != 0) return BMI_ERR_UNSUPPORTED;
    if (a.N <= 0 || a.in_mod <= 0 || a.B <= 0) return BMI_ERR_INVALID;
    const long total = (long)a.N * a.HW * (a.C >> 3);
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    static const int share = [] { const char* v = std::getenv("BMI_MASK_SHARE"); return v ? std::atoi(v) : 1; }();
    const int lb = a.site.log2_bits;
    const long sample_elems = (long)a.B * a.HW * a.C;
    if (share && a.site.kind == BMI_SITE_ELEMENTWISE && lb >= 1 && lb <= 3 && a.N % a.B == 0 && (a.in_mod == a.B || a.in_mod == a.N) &&
        sample_elems % (64L * (128 >> lb)) == 0) {
        const long n_sb = sample_elems / (64L * (128 >> lb)) * (a.N / a.B);
        long wblocks = (n_sb + 3) / 4;                 // 4 waves per block, one super-block per wave and iteration
        if (wblocks > 256 * 16) wblocks = 256 * 16;
        const dim3 g((unsigned)wblocks), b(256);
        static const int lean = [] { const char* v = std::getenv("BMI_MASK_LEAN"); return v ? std::atoi(v) : 1; }();
        if (lean && lb == 1 && !a.bias_post && !a.relu) {
            // samples per work item (deterministic input): as many as still leave ~2 items per wave slot of the chip
            EltArgs al = a;
            const long sbs = sample_elems / 4096, T = a.N / a.B;
            long chunks = (16384 + sbs - 1) / sbs;
            if (chunks > T) chunks = T;
            al.tchunk = (int)((T + chunks - 1) / chunks);
            const long items = sbs * (a.in_mod == a.B ? (T + al.tchunk - 1) / al.tchunk : T);
            long lblocks = (items + 3) / 4;
            if (lblocks > 256 * 16) lblocks = 256 * 16;
            const dim3 g((unsigned)lblocks);
            const EltArgs& a = al;
            if (a.bf16) hipLaunchKernelGGL((mask_apply_lb1_kernel<true>), g, b, 0, s, a);
            else hipLaunchKernelGGL((mask_apply_lb1_kernel<false>), g, b, 0, s, a);
            BMI_CHECK_LAUNCH();
            return BMI_OK;
        }
        if (a.bf16) {
            if (lb == 1) hipLaunchKernelGGL((mask_apply_shared_kernel<1, true>), g, b, 0, s, a);
            else if (lb == 2) hipLaunchKernelGGL((mask_apply_shared_kernel<2, true>), g, b, 0, s, a);
            else hipLaunchKernelGGL((mask_apply_shared_kernel<3, true>), g, b, 0, s, a);
        } else {
            if (lb == 1) hipLaunchKernelGGL((mask_apply_shared_kernel<1, false>), g, b, 0, s, a);
            else if (lb == 2) hipLaunchKernelGGL((mask_apply_shared_kernel<2, false>), g, b, 0, s, a);
            else hipLaunchKernelGGL((mask_apply_shared_kernel<3, false>), g, b, 0, s, a);
        }
        BMI_CHECK_LAUNCH();
        return BMI_OK;
    }
    if (a.bf16) hipLaunchKernelGGL(mask_apply_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(mask_apply_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}

// ---------------------------------------------------------------------------------------------
// Keep bits of an elementwise site for a folded batch: byte g of sample t_local holds the keep flags
// of elements 8g..8g+7 (bit e = element 8g+e).  16x smaller than the masked activation: consumers that
// stage their input through registers (conv_igemm) read the un-expanded deterministic tensor from
// L2 plus these bits instead of a materialised x * mask copy from HBM.
__global__ __launch_bounds__(256) void mask_bits_kernel(uint8_t* __restrict__ bits, long groups_per_sample, int tc, int t0,
                                                        SiteArgs s) {
    const long total4 = (groups_per_sample * tc + 3) / 4;   // 4 groups (one dword of bits) per thread
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        uint32_t word = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long gi = i * 4 + q;
            if (gi >= groups_per_sample * tc) break;
            const long tl = gi / groups_per_sample;
            const uint64_t g = (uint64_t)(gi - tl * groups_per_sample);
            const uint32_t b = site_keep8(s, g << 3, (uint32_t)(t0 + tl));
            word |= b << (8 * q);
        }
        if (i * 4 + 3 < groups_per_sample * tc) *(uint32_t*)(bits + i * 4) = word;
        else for (int q = 0; q < 4 && i * 4 + q < groups_per_sample * tc; ++q) bits[i * 4 + q] = (uint8_t)(word >> (8 * q));
    }
}

// out = x * scale rounded once to the activation type: what mask_apply stores for a kept element (lazy sites: the consumers AND
// this copy of the B deterministic images with the keep bits).
// PLANAR (w, c > 0): the copy is stored in the layout its stride-2 consumers DMA whole lines from (kernels.h, lazy_planar_off):
// [image][c / 32][y][x & 1][x >> 1][c & 31] instead of NHWC.
template <bool BF>
__global__ __launch_bounds__(256) void scale_copy_kernel(const _Float16* __restrict__ in, _Float16* __restrict__ out, long n8, float scale,
                                                         int hw, int w, int c) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        long src = i * 8;
        if (w > 0) {                                     // thread i = 8 elements at OUTPUT offset 8 i: (image, plane, position, 8-channel group)
            const long per_img = (long)hw * c;
            const long img = (i * 8) / per_img;
            const long rem = i * 8 - img * per_img;
            const int plane = (int)(rem / ((long)hw * 32));
            const long r2 = rem - (long)plane * hw * 32;
            const int pos = (int)(r2 >> 5), c8 = (int)(r2 & 31);
            const int y = pos / w, xx = pos - y * w, hwid = w >> 1;
            const int x = xx < hwid ? 2 * xx : 2 * (xx - hwid) + 1;
            src = img * per_img + ((long)y * w + x) * c + plane * 32 + c8;
        }
        const half8 x = *(const half8*)(in + src);
        half8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = a16_from_f32<BF>(a16_to_f32<BF>(x[e]) * scale);
        *(half8*)(out + i * 8) = r;
    }
}

int launch_scale_copy(const _Float16* in, _Float16* out, long n, float scale, int bf16, hipStream_t s, int planar_hw, int planar_w, int planar_c) {
    if (n <= 0 || n % 8 != 0) return BMI_ERR_INVALID;
    if (planar_w > 0 && (planar_c % 32 != 0 || (planar_w & 1) || planar_hw % planar_w != 0 || n % ((long)planar_hw * planar_c) != 0)) return BMI_ERR_INVALID;
    long blocks = (n / 8 + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (bf16) hipLaunchKernelGGL(scale_copy_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, in, out, n / 8, scale, planar_hw, planar_w, planar_c);
    else hipLaunchKernelGGL(scale_copy_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, in, out, n / 8, scale, planar_hw, planar_w, planar_c);
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}

// The same bits from ONE Philox call per 128 >> LB elements (the kernel above draws a call per byte: 8 / 4 / 2 times the calls
// at 2 / 4 / 8 bits per element — 0.70 ms for the 1.6 G elements of the headline's first site, as long as writing the masked
// tensor itself): a thread takes a call and stores its 8 / 4 / 2 bytes of keep flags.
// PLANAR (LB = 1, w > 0): the flags of element (c, y, x) of an image sit at bit lazy_planar_off(c, y, x) of the image's bits — the
// permutation of the scaled copy (scale_copy_kernel): a call's 64 channels are two dwords, one per 32-channel plane.
template <int LB>
__global__ __launch_bounds__(256) void mask_bits_call_kernel(uint8_t* __restrict__ bits, long calls_per_sample, int tc, int t0, SiteArgs s,
                                                             int hw, int w, int c) {
    constexpr int EPC = 128 >> LB, GPC = EPC / 8;
    const long total = calls_per_sample * tc;
    const long imgs_per_sample = (GPC == 8 && w > 0) ? calls_per_sample / ((long)hw * (c >> 6)) : 0;      // (planar) B
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long tl = i / calls_per_sample;
        uint64_t elem0 = (uint64_t)(i - tl * calls_per_sample) * EPC;
        long pl_word = 0;                  // planar: dword index of this call's first 32-channel plane
        if (GPC == 8 && w > 0) {
            // planar: thread i = (sample, image, 64-channel group, POSITION in the plane): consecutive threads store consecutive dwords;
            // the position's pixel is (y, x) with the even columns of a row in front of the odd ones.  hw and w are powers of two
            // (launcher): shifts only — this kernel is bound by its integer instructions (Philox), 64-bit divisions showed (0.12 -> 0.21 ms)
            const int lhw = __builtin_ctz((unsigned)hw), lw = __builtin_ctz((unsigned)w);
            const unsigned cpp = (unsigned)c >> 6;
            const unsigned r = (unsigned)(i - tl * calls_per_sample);       // (b * cpp + cgrp) * hw + pos inside the sample
            const unsigned pos = r & (unsigned)(hw - 1), bc = r >> lhw;
            const unsigned b = cpp == 1 ? bc : bc / cpp, cgrp = bc - b * cpp;
            const unsigned y = pos >> lw, xx = pos & (unsigned)(w - 1), hwid = (unsigned)w >> 1;
            const unsigned x = xx < hwid ? 2 * xx : 2 * (xx - hwid) + 1;
            elem0 = ((((uint64_t)b << lhw) + (y << lw) + x) * cpp + cgrp) << 6;
            const long img = tl * imgs_per_sample + b;                      // folded image index
            pl_word = (img * (long)cpp * 2 + 2 * cgrp) * hw + pos;
        }
        uint64_t out = 0;
        if (!s.drop_all) {
            const philox4 r = philox_site_call(s, elem0, (uint32_t)(t0 + tl));
            if constexpr (LB == 1) {
                // 2 bits per element: word w holds elements 16 w .. 16 w + 15, field e at bits 2e.  field >= thresh on all 16 fields at once
                // (b0 / b1 = the even / odd bits: b1 | b0 for thresh 1, b1 for 2, b1 & b0 for 3), then the even bits are packed into 16
                const uint32_t use_or = s.thresh == 1 ? 0xffffffffu : 0u, use_and = s.thresh == 3 ? 0xffffffffu : 0u;
                const uint32_t keep_all = s.thresh == 0 ? 0x55555555u : 0u;     // p = 0: every field >= 0 (thresh 1..3 below)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const uint32_t f = r.w[w], b0 = f & 0x55555555u, b1 = (f >> 1) & 0x55555555u;
                    uint32_t x = (((b1 | (b0 & use_or)) & (b0 | ~use_and)) | keep_all) & 0x55555555u;
                    x = (x | (x >> 1)) & 0x33333333u;
                    x = (x | (x >> 2)) & 0x0f0f0f0fu;
                    x = (x | (x >> 4)) & 0x00ff00ffu;
                    x = (x | (x >> 8)) & 0x0000ffffu;
                    out |= (uint64_t)x << (16 * w);
                }
            } else {
#pragma unroll
                for (int q = 0; q < GPC; ++q) out |= (uint64_t)philox_keep8(r, (uint32_t)elem0 + 8u * q, LB, s.thresh) << (8 * q);
            }
        }
        if constexpr (GPC == 8) {
            if (w > 0) {
                ((uint32_t*)bits)[pl_word] = (uint32_t)out;
                ((uint32_t*)bits)[pl_word + hw] = (uint32_t)(out >> 32);
            } else {
                *(uint64_t*)(bits + i * 8) = out;
            }
        }
        else if constexpr (GPC == 4) *(uint32_t*)(bits + i * 4) = (uint32_t)out;
        else *(uint16_t*)(bits + i * 2) = (uint16_t)out;
    }
}

int launch_mask_bits(uint8_t* bits, int n, int hw, int c, const SiteArgs& site, int batch, int t0, hipStream_t s, int planar_w) {
    if (c % 8 != 0 || site.kind != BMI_SITE_ELEMENTWISE) return BMI_ERR_UNSUPPORTED;
    if (planar_w > 0 && (site.log2_bits != 1 || c % 64 != 0 || planar_w < 2 || (planar_w & (planar_w - 1)) || (hw & (hw - 1)) || hw % planar_w != 0))
        return BMI_ERR_UNSUPPORTED;
    if (n <= 0 || batch <= 0 || n % batch != 0) return BMI_ERR_INVALID;
    const long gps = (long)batch * hw * (c / 8);
    const int tc = n / batch;
    const int lb = site.log2_bits;
    if (lb >= 1 && lb <= 3 && ((long)batch * hw * c) % (128 >> lb) == 0) {
        const long cps = (long)batch * hw * c / (128 >> lb);
        long cblocks = (cps * tc + 255) / 256;
        if (cblocks > 256 * 32) cblocks = 256 * 32;
        const dim3 g((unsigned)cblocks), b(256);
        if (lb == 1) hipLaunchKernelGGL(mask_bits_call_kernel<1>, g, b, 0, s, bits, cps, tc, t0, site, hw, planar_w, c);
        else if (lb == 2) hipLaunchKernelGGL(mask_bits_call_kernel<2>, g, b, 0, s, bits, cps, tc, t0, site, hw, 0, c);
        else hipLaunchKernelGGL(mask_bits_call_kernel<3>, g, b, 0, s, bits, cps, tc, t0, site, hw, 0, c);
        BMI_CHECK_LAUNCH();
        return BMI_OK;
    }
    if (planar_w > 0) return BMI_ERR_UNSUPPORTED;
    long blocks = ((gps * tc + 3) / 4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(mask_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, s, bits, gps, tc, t0, site);
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}

// ---------------------------------------------------------------------------------------------
template <bool BF>
__global__ __launch_bounds__(256) void maxpool2_kernel(const _Float16* in, _Float16* out, int N, int H, int W, int C) {
    const int cg = C >> 3, Ho = H >> 1, Wo = W >> 1;
    const long total = (long)N * Ho * Wo * cg;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cg) * 8;
        long q = i / cg;
        const int ox = (int)(q % Wo); q /= Wo;
        const int oy = (int)(q % Ho);
        const int n = (int)(q / Ho);
        const _Float16* p = in + (((size_t)n * H + 2 * oy) * W + 2 * ox) * C + c8;
        const half8 a0 = *(const half8*)p, a1 = *(const half8*)(p + C);
        const half8 a2 = *(const half8*)(p + (size_t)W * C), a3 = *(const half8*)(p + (size_t)W * C + C);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float m = fmaxf(fmaxf(a16_to_f32<BF>(a0[e]), a16_to_f32<BF>(a1[e])), fmaxf(a16_to_f32<BF>(a2[e]), a16_to_f32<BF>(a3[e])));
            o[e] = a16_from_f32<BF>(m);
        }
        *(half8*)(out + (((size_t)n * Ho + oy) * Wo + ox) * C + c8) = o;
    }
}

int launch_maxpool2(const _Float16* in, _Float16* out, int n, int h, int w, int c, int bf16, hipStream_t s) {
    if (c % 8 != 0 || (h & 1) || (w & 1)) return BMI_ERR_UNSUPPORTED;
    const long total = (long)n * (h / 2) * (w / 2) * (c / 8);
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks <= 0) return BMI_ERR_INVALID;
    if (bf16) hipLaunchKernelGGL(maxpool2_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, in, out, n, h, w, c);
    else hipLaunchKernelGGL(maxpool2_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, in, out, n, h, w, c);
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}

// ---------------------------------------------------------------------------------------------
// Direct convolution for the 3-channel network input; weights [Cout][k][k][Cin] fp32 staged (transposed) in LDS.
// fp32 math (same fmaf order over the taps as before: identical results), fp16 NHWC store.
#define STEM_MAX_W 4096  // floats of weight in LDS (64 x 3x3x3 = 1728)
// One thread = one output pixel x 32 output channels (lane = consecutive pixels: the fp32 NCHW reads and the 64-byte NHWC
// stores are coalesced; a tap's 32 weights are an LDS broadcast of 8 x ds_read_b128).  The first version mapped 8 lanes to
// one pixel (8 channels each): every wave-load fetched 8 distinct words and the kernel sat at 80 us for 0.9 GFLOP.
// FAST: three input channels and whole 32-channel blocks (every model of the path): no per-quad channel guards, the three
// loads of a tap issued together.
// F32OUT (the exact engine, BMI_DTYPE_F32): `out` holds fp32.
// PAIROUT (the split engines): `out` is a pair32 tensor (conv_epilogue.h) of fp16 (BF = false) or bf16 halves; a thread's 32 channels are one block.
template <bool BF, bool FAST, bool F32OUT = false, bool PAIROUT = false>
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ scale, const float* __restrict__ bias,
                                                        _Float16* __restrict__ out, int N, int Cin_, int H, int W, int Cout,
                                                        int ks, int stride, int pad, int Ho, int Wo, int relu) {
    const int Cin = FAST ? 3 : Cin_;
    __shared__ __attribute__((aligned(16))) float wl[STEM_MAX_W];      // transposed: [tap][Cout]
    const int kvol = ks * ks * Cin;
    for (int i = threadIdx.x; i < Cout * kvol; i += blockDim.x) { const int c = i / kvol, t = i - c * kvol; wl[t * Cout + c] = w[i]; }
    __syncthreads();
    const int c0 = blockIdx.y * 32;                                    // this block's 32 output channels (uniform: LDS broadcast)
    const int nc = FAST ? 32 : min(32, Cout - c0);                     // multiple of 8
    const long total = (long)N * Ho * Wo;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    long q = i;
    const int ox = (int)(q % Wo); q /= Wo;
    const int oy = (int)(q % Ho);
    const int n = (int)(q / Ho);
    float acc[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) acc[e] = 0.f;
    for (int ky = 0; ky < ks; ++ky) {
        const int iy = oy * stride - pad + ky;
        const bool oky = (unsigned)iy < (unsigned)H;
#pragma unroll 1
        for (int kx = 0; kx < ks; ++kx) {
            const int ix = ox * stride - pad + kx;
            const bool ok = oky && (unsigned)ix < (unsigned)W;
            float xin[3] = {0.f, 0.f, 0.f};
            if constexpr (FAST) {
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) xin[ci] = ok ? x[(((size_t)n * 3 + ci) * H + iy) * W + ix] : 0.f;
            }
#pragma unroll
            for (int ci = 0; ci < (FAST ? 3 : Cin); ++ci) {
                const float xv = FAST ? xin[ci] : (ok ? x[(((size_t)n * Cin + ci) * H + iy) * W + ix] : 0.f);
                const float* wr = wl + ((ky * ks + kx) * Cin + ci) * Cout + c0;
#pragma unroll
                for (int e4 = 0; e4 < 8; ++e4) {
                    if (FAST || 4 * e4 < nc) {
                        const float4 w4 = *(const float4*)(wr + 4 * e4);
                        acc[4 * e4] = fmaf(xv, w4.x, acc[4 * e4]); acc[4 * e4 + 1] = fmaf(xv, w4.y, acc[4 * e4 + 1]);
                        acc[4 * e4 + 2] = fmaf(xv, w4.z, acc[4 * e4 + 2]); acc[4 * e4 + 3] = fmaf(xv, w4.w, acc[4 * e4 + 3]);
                    }
                }
            }
        }
    }
    const size_t o_off = (((size_t)n * Ho + oy) * Wo + ox) * Cout + c0;
    _Float16* op = out + o_off;
    if constexpr (PAIROUT) op = out + pair32_off(((size_t)n * Ho + oy) * Wo + ox, Cout, c0);
#pragma unroll
    for (int g8 = 0; g8 < 4; ++g8) {
        if (8 * g8 < nc) {
            half8 o;
            float of[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = acc[8 * g8 + e];
                if (scale) v *= scale[c0 + 8 * g8 + e];
                if (bias) v += bias[c0 + 8 * g8 + e];
                if (relu) v = fmaxf(v, 0.f);
                if constexpr (F32OUT || PAIROUT) of[e] = v;
                else o[e] = a16_from_f32<BF>(v);
            }
            if constexpr (PAIROUT) {
                pair_encode<BF, 8>(op + 8 * g8, of);
            } else if constexpr (F32OUT) {
                float* fp = (float*)out + o_off + 8 * g8;
                *(float4*)fp = make_float4(of[0], of[1], of[2], of[3]);
                *(float4*)(fp + 4) = make_float4(of[4], of[5], of[6], of[7]);
            } else {
                *(half8*)(op + 8 * g8) = o;
            }
        }
    }
}

// dt: BMI_DTYPE_* of `out` (F16 / BF16: 16-bit NHWC; F32: fp32 NHWC)
int launch_stem_conv(const float* x, const float* w, const float* scale, const float* bias, _Float16* out, int n,
                     int cin, int h, int wdt, int cout, int ksize, int stride, int pad, int relu, int dt, hipStream_t s) {
    const int bf16 = dt == BMI_DTYPE_BF16;
    if (cout % 8 != 0 || cout * ksize * ksize * cin > STEM_MAX_W) return BMI_ERR_UNSUPPORTED;
    if (n <= 0) return BMI_ERR_INVALID;
    const int ho = (h + 2 * pad - ksize) / stride + 1, wo = (wdt + 2 * pad - ksize) / stride + 1;
    const long total = (long)n * ho * wo;
    const dim3 grid((unsigned)((total + 255) / 256), (unsigned)((cout + 31) / 32)), block(256);
#define STEM_LAUNCH(BF_, F_) hipLaunchKernelGGL((stem_conv_kernel<BF_, F_>), grid, block, 0, s, x, w, scale, bias, out, n, cin, h, wdt, cout, ksize, stride, pad, ho, wo, relu)
    if (dt == BMI_DTYPE_F16X2 || dt == BMI_DTYPE_BF16X3) {      // pair32 output: whole 32-channel blocks
        if (cin != 3 || cout % 32 != 0) return BMI_ERR_UNSUPPORTED;
        if (dt == BMI_DTYPE_BF16X3) hipLaunchKernelGGL((stem_conv_kernel<true, true, false, true>), grid, block, 0, s, x, w, scale, bias, out, n, cin, h, wdt, cout, ksize, stride, pad, ho, wo, relu);
        else hipLaunchKernelGGL((stem_conv_kernel<false, true, false, true>), grid, block, 0, s, x, w, scale, bias, out, n, cin, h, wdt, cout, ksize, stride, pad, ho, wo, relu);
    } else if (dt == BMI_DTYPE_F32) {
        if (cin == 3 && cout % 32 == 0) hipLaunchKernelGGL((stem_conv_kernel<false, true, true>), grid, block, 0, s, x, w, scale, bias, out, n, cin, h, wdt, cout, ksize, stride, pad, ho, wo, relu);
        else hipLaunchKernelGGL((stem_conv_kernel<false, false, true>), grid, block, 0, s, x, w, scale, bias, out, n, cin, h, wdt, cout, ksize, stride, pad, ho, wo, relu);
    } else if (cin == 3 && cout % 32 == 0) { if (bf16) STEM_LAUNCH(true, true); else STEM_LAUNCH(false, true); }
    else { if (bf16) STEM_LAUNCH(true, false); else STEM_LAUNCH(false, false); }
#undef STEM_LAUNCH
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}

// ---------------------------------------------------------------------------------------------
// Dynamic early exit, the per-stage confidence test + compaction of the still-active images (what the reference models post
// hoc: FullAnalysis.confidence_exiting / is_confident, SA/train/results_analyzer.py:606-630, :725-733): among the `bc`
// active images (list `in`, identity when null) those whose exit-e confidence max_c mean_t p[e][b][c] exceeds the
// threshold get exit_of[b] = e and leave; the others are written, order preserved, to `out` (`in` != `out`) and counted.
__global__ __launch_bounds__(256) void exit_decide_kernel(const double* __restrict__ S1e, int C, double inv_t, double thr,
                                                          const int* __restrict__ in, int bc, int* __restrict__ out,
                                                          int* __restrict__ count, int* __restrict__ exit_of, int e) {
    __shared__ int scan[256];
    __shared__ int base_s;
    const int tid = threadIdx.x;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int base = 0; base < bc; base += 256) {
        const int i = base + tid;
        int keep = 0, b = -1;
        if (i < bc) {
            b = in ? in[i] : i;
            double mx = 0.0;
            for (int c = 0; c < C; ++c) mx = fmax(mx, S1e[(size_t)b * C + c]);
            if (mx * inv_t > thr) exit_of[b] = e;
            else keep = 1;
        }
        scan[tid] = keep;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int v = tid >= off ? scan[tid - off] : 0;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        if (keep) out[base_s + scan[tid] - 1] = b;
        __syncthreads();
        if (tid == 255) base_s += scan[255];
        __syncthreads();
    }
    if (tid == 0) *count = base_s;
}

int launch_exit_decide(const double* S1e, int C, int t_total, double thr, const int* in, int bc, int* out, int* count,
                       int* exit_of, int e, hipStream_t s) {
    if (!S1e || !out || !count || !exit_of || bc <= 0 || C <= 0 || t_total <= 0 || in == out) return BMI_ERR_INVALID;
    hipLaunchKernelGGL(exit_decide_kernel, dim3(1), dim3(256), 0, s, S1e, C, 1.0 / t_total, thr, in, bc, out, count, exit_of, e);
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}

__global__ void expand_rows_kernel(const int* __restrict__ active, int bc, int batch, int total, int* __restrict__ rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int tl = i / bc;
    rows[i] = tl * batch + active[i - tl * bc];
}

int launch_expand_rows(const int* active, int bc, int batch, int tc, int* rows, hipStream_t s) {
    if (!active || !rows || bc <= 0 || batch <= 0 || tc <= 0) return BMI_ERR_INVALID;
    const int total = bc * tc;
    hipLaunchKernelGGL(expand_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, active, bc, batch, total, rows);
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}

__global__ void fill_int_kernel(int* p, int n, int v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void mask_permute_kernel(const float* __restrict__ src, float* __restrict__ dst, int m, int c, int cnt0, int stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * c) return;
    const int r = i / c, k = i - r * c;
    dst[i] = src[(size_t)((cnt0 + (long)r * stride) % m) * c + k];
}

int launch_mask_permute(const float* src, float* dst, int m, int c, int cnt0, int stride, hipStream_t s) {
    if (!src || !dst || m <= 0 || c <= 0 || stride < 1 || cnt0 < 0) return BMI_ERR_INVALID;
    hipLaunchKernelGGL(mask_permute_kernel, dim3((unsigned)((m * c + 255) / 256)), dim3(256), 0, s, src, dst, m, c, cnt0, stride);
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}

int launch_fill_int(int* p, int n, int v, hipStream_t s) {
    if (!p || n <= 0) return BMI_ERR_INVALID;
    hipLaunchKernelGGL(fill_int_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, v);
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}

// CHECK: also counts the elements whose sums are not finite (an fp16 activation that saturated at 65 504 turns into inf, then NaN in the
// softmax) into *nonfinite — one ballot per wavefront, one atomic per wavefront that saw any.
template <bool CHECK>
__global__ void finalize_kernel(long n, double inv_t, const double* S1, const double* S2, const double* SL, double* mean,
                                double* var, double* lm, int* nonfinite) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    if (i < n) {
        const double s1 = S1[i], s2 = S2[i], sl = SL[i];
        const double m = s1 * inv_t;
        const double v = s2 * inv_t - m * m;
        mean[i] = m;
        var[i] = v > 0 ? v : (v == v ? 0 : v);     // (a NaN stays a NaN: the clamp must not hide it)
        lm[i] = sl * inv_t;
        if (CHECK) bad = !(__builtin_isfinite(s1) && __builtin_isfinite(s2) && __builtin_isfinite(sl));
    }
    if (CHECK) {
        const unsigned long long b = __ballot(bad);
        if (b && (threadIdx.x & 63) == 0) atomicAdd(nonfinite, __popcll(b));
    }
}

int launch_finalize(int64_t n, int t_total, const double* S1, const double* S2, const double* SL, double* mean,
                    double* var, double* lm, int* nonfinite, hipStream_t s) {
    if (n <= 0 || t_total <= 0) return BMI_ERR_INVALID;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (nonfinite) hipLaunchKernelGGL(finalize_kernel<true>, grid, block, 0, s, (long)n, 1.0 / t_total, S1, S2, SL, mean, var, lm, nonfinite);
    else hipLaunchKernelGGL(finalize_kernel<false>, grid, block, 0, s, (long)n, 1.0 / t_total, S1, S2, SL, mean, var, lm, nonfinite);
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}

// ---------------------------------------------------------------------------------------------
__global__ void philox_mask_kernel(uint8_t* keep, long n, SiteArgs s, int t) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;     // one 8-element group per thread
    if (g * 8 >= n) return;
    const uint32_t bits = site_keep8(s, (uint64_t)g << 3, (uint32_t)t);
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (g * 8 + e < n) keep[g * 8 + e] = (bits >> e) & 1u;
}

int launch_philox_mask(uint8_t* keep, int64_t n, uint64_t seed, int site, int t, float p, hipStream_t s) {
    if (n <= 0) return BMI_ERR_INVALID;
    bmi_site bs;
    bs.kind = BMI_SITE_ELEMENTWISE; bs.site_id = site; bs.p = p; bs.num_masks = 0; bs.masks = nullptr;
    const SiteArgs sa = resolve_site(&bs, seed, 0);
    const long groups = (n + 7) / 8;
    hipLaunchKernelGGL(philox_mask_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, keep, (long)n, sa, t);
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}
