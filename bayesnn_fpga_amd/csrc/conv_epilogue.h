// Fused conv epilogue shared by the implicit-GEMM and the patch kernels: folded-BN scale/bias,
// residual add, ReLU, stochastic site, fp16 NHWC store — for ONE accumulator quad (4 consecutive
// output channels of one pixel, the (reg & 3) registers of v_mfma_f32_32x32x16 with channels on
// the row axis).  Reference semantics: BasicBlock.forward SA/models/resnet18/resnet18.py:32-48,
// MCDropout :207-210, Masksembles2D SA/utils.py:165-169.
#pragma once
#include "kernels.h"

typedef _Float16 half4 __attribute__((ext_vector_type(4)));

// Activation / weight element type.  Everything 16-bit is STORED as `_Float16` lanes (an opaque 16-bit container: the
// pointer types, LDS images, DMA and fragment reads never look inside); BF = true means the bits are bfloat16.  Only
// the two ends differ: the MFMA instruction (v_mfma_*_bf16) and the fp32 <-> 16-bit conversions below.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
template <bool BF>
__device__ __forceinline__ float a16_to_f32(_Float16 h) {
    if constexpr (BF) return (float)__builtin_bit_cast(__bf16, h);
    else return (float)h;
}
template <bool BF>
__device__ __forceinline__ _Float16 a16_from_f32(float v) {
    if constexpr (BF) return __builtin_bit_cast(_Float16, (__bf16)v);   // v_cvt_pk_bf16_f32: round to nearest even
    else return (_Float16)v;
}
typedef float f32x4_m __attribute__((ext_vector_type(4)));
typedef float f32x16_m __attribute__((ext_vector_type(16)));

// ---- "pair32": how the SPLIT engines (BMI_DTYPE_F16X2 / BF16X3) keep an activation tensor in the workspace -------------------------
// Every element is a 16-bit head and tail, v = hi + lo (hi = rn16(v), lo = rn16(v - hi)) — the operand form conv_split.hip multiplies —
// stored per pixel in 32-channel blocks [hi x 32 | lo x 32]: element (pixel, c) has its head at 16-bit index
//     pixel * 2 C + (c >> 5) * 64 + (c & 31)          and its tail 32 further,
// 4 bytes per element like fp32, and a K-step's operand (one pixel, 32 channels, both halves) is ONE contiguous 128-byte line: the conv
// fetches it by LDS-DMA, nothing is split in the consumer (the producer encodes once what nine taps and every channel tile would
// otherwise split again).  C % 32 == 0.
__host__ __device__ inline size_t pair32_off(size_t pixel, int C, int c) { return pixel * 2 * (size_t)C + (size_t)(c >> 5) * 64 + (c & 31); }
typedef _Float16 half4_p __attribute__((ext_vector_type(4)));
template <bool BF, int N>     // N = 4 | 8 consecutive channels inside one 32-block; p = address of the first head
__device__ __forceinline__ void pair_decode(const _Float16* p, float v[N]) {
    typedef _Float16 hv __attribute__((ext_vector_type(N)));
    const hv hi = *(const hv*)p, lo = *(const hv*)(p + 32);
#pragma unroll
    for (int e = 0; e < N; ++e) v[e] = a16_to_f32<BF>(hi[e]) + a16_to_f32<BF>(lo[e]);
}
template <bool BF, int N>
__device__ __forceinline__ void pair_encode(_Float16* p, const float v[N]) {
    typedef _Float16 hv __attribute__((ext_vector_type(N)));
    hv hi, lo;
#pragma unroll
    for (int e = 0; e < N; ++e) {
        hi[e] = a16_from_f32<BF>(v[e]);
        lo[e] = a16_from_f32<BF>(v[e] - a16_to_f32<BF>(hi[e]));
    }
    *(hv*)p = hi;
    *(hv*)(p + 32) = lo;
}
template <bool BF>
__device__ __forceinline__ f32x4_m mfma_16x16x32(half8_t a, half8_t b, f32x4_m c) {
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
template <bool BF>
__device__ __forceinline__ f32x16_m mfma_32x32x16(half8_t a, half8_t b, f32x16_m c) {
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// XCD-aware tile mapping.  Workgroups are dealt round-robin over the 8 XCDs (block b -> XCD b % 8,
// each with its own L2), so the n_ct channel tiles that share ONE pixel tile must get block ids
// that are congruent mod 8, or every XCD fetches that pixel tile's input again from beyond its L2
// (measured: FETCH_SIZE 3.3-3.8x the algorithmic input bytes at Cout = 512).  Within a group of
// 8 * n_ct consecutive blocks, block j takes pixel tile (j % 8) of the group and channel tile j / 8.
// Placement only changes speed, never results; the tail group falls back to the plain order.
//
// `cs` (1, 2 or 4; n_ct % cs == 0) splits the channel tiles into cs classes and gives each class its own XCDs: XCD x only
// ever streams the weights of class x % cs.  With cs = 1 every XCD streams ALL the weights of the conv; at 512 -> 512
// channels that is 4.7 MB of fp16 through a 4 MB L2 in a cyclic pattern, and the workgroups re-fetch their 1.18 MB
// slices from beyond L2 (measured round 1: FETCH_SIZE 2.7 GB per launch against 0.82 GB algorithmic on the 4x4-map
// convs).  With cs = 2 an XCD's weights are 2.4 MB (resident) and a pixel tile's input is fetched by two XCDs instead
// of one (the second fetch is a hit in the memory-side cache).  The launcher picks cs from the weight bytes.
__device__ __forceinline__ void xcd_tile_map(int bid, int n_pt, int n_ct, int& ptile, int& ctile, int cs = 1) {
    const int group = 8 * n_ct;
    const int full = (n_pt / 8) * group;
    if (bid < full) {
        const int g = bid / group, j = bid - g * group;
        const int x = j & 7, s = j >> 3;          // XCD, position in the XCD's share of the group
        const int cpc = n_ct / cs;                // channel tiles per class
        const int cls = x % cs, xg = x / cs;
        const int sc = s % cpc, sp = s / cpc;
        ptile = g * 8 + xg * cs + sp;
        ctile = cls * cpc + sc;
    } else {
        const int t = bid - full;
        ptile = (n_pt / 8) * 8 + t / n_ct;
        ctile = t % n_ct;
    }
}

// Tile order of a launch that reads a DETERMINISTIC input of B images through a lazy site (keep bits per sample, engine.hip).  In the
// plain order tile -> pixel m = (t B + b) Ho Wo + ..: neighbours in time are different IMAGES of one sample, and the T samples' reads of
// one image's activations lie B Ho Wo / tile workgroups apart — each of them a miss in the XCD's 4 MB L2 (rocprofv3, ResNet-50's 1x1
// readers: 5.7 GB fetched per launch for 0.13 GB of activations + 0.5 GB of bits; its 3x3 stride-2 reader 19.6 GB).  Here the tiles are
// numbered sample-MINOR, id = ((b PT + pxt) T + t) n_ct + ct (PT pixel tiles per image, n_ct channel tiles), and XCD x (blocks = x mod 8)
// takes the contiguous ids [x L, (x + 1) L), L = ceil(total / 8), in block order: the T n_ct workgroups that read one activation tile run
// back to back on one XCD and all but the first find it in L2.  Grid = 8 L blocks (lazy_tile_grid); false = no tile for this block.
// Placement only: results do not change.
__device__ __forceinline__ bool lazy_tile_map(int bid, int B, int T, int PT, int n_ct, int& ptile, int& ctile) {
    const unsigned total = (unsigned)B * PT * T * n_ct;          // (launcher: < 2^31)
    const unsigned L = (total + 7u) >> 3;
    const unsigned x = (unsigned)bid & 7u, s = (unsigned)bid >> 3;
    const unsigned id = x * L + s;
    if (s >= L || id >= total) return false;
    unsigned r = id;
    ctile = (int)(r % (unsigned)n_ct); r /= (unsigned)n_ct;
    const unsigned t = r % (unsigned)T; r /= (unsigned)T;
    const unsigned pxt = r % (unsigned)PT, b = r / (unsigned)PT;
    ptile = (int)((t * B + b) * PT + pxt);
    return true;
}
inline long lazy_tile_grid(long total) { return 8 * ((total + 7) / 8); }

// Dynamic early exit: compact image index of this launch -> row of the tensors / Philox image index (see ConvArgs::imap).
// IMAP is a KERNEL TEMPLATE PARAMETER: the ordinary instantiations (IMAP = false) contain no trace of it.  As a run-time
// `a.imap ? a.imap[n] : n` inside the epilogue's load loops it made hipcc wait vmcnt(0) around every conditional load and
// serialised the residual prefetch (+24 % on the general-epilogue wide kernel with no dynamic exit in sight), and a
// division by the active-image count instead of the row table cost 16-24 spilled VGPRs.
template <bool IMAP>
__device__ __forceinline__ int map_image(const ConvArgs& a, int n) {
    if constexpr (IMAP) return n < a.N ? a.imap[n] : n;
    else return n;
}

struct PixelCtx {
    size_t out_off;        // element offset of this pixel's channel 0 in the output tensor
    const _Float16* resp;  // residual row or nullptr
    const float* mrow;     // Masksembles row or nullptr
    int e_pix;             // pixel index inside one Monte-Carlo sample: (b*Ho + y)*Wo + x
    int b;                 // image index inside the sample
    int t;                 // global sample index
};

__device__ __forceinline__ PixelCtx make_pixel_ctx(const ConvArgs& a, int n, int rem /* y*Wo + x */) {
    const int HoWo = a.Ho * a.Wo;
    PixelCtx p;
    p.out_off = ((size_t)n * HoWo + rem) * a.Cout;
    p.resp = a.res ? a.res + ((size_t)(n % a.res_mod) * HoWo + rem) * a.Cout : nullptr;
    const int tl = n / a.B;
    p.b = n - tl * a.B;
    p.t = a.t0 + tl;
    p.e_pix = p.b * HoWo + rem;
    p.mrow = a.site.kind == BMI_SITE_MASKSEMBLE
                 ? a.site.masks + (size_t)((a.site.cnt0 + p.t) % a.site.num_masks) * a.Cout
                 : nullptr;
    return p;
}

// multipliers of the stochastic site for one accumulator quad (4 consecutive channels c4.. of pixel p)
__device__ __forceinline__ void site_mult4(const ConvArgs& a, const PixelCtx& p, int c4, float m[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) m[e] = 1.f;
    if (a.site.kind == BMI_SITE_ELEMENTWISE || a.site.kind == BMI_SITE_CHANNEL) {
        const int row = a.site.kind == BMI_SITE_ELEMENTWISE ? p.e_pix : p.b;   // (one multiply: a select between two
        const uint64_t elem = (uint64_t)row * a.Cout + c4;                      //  products went through scratch)
        const uint32_t keep = site_keep8(a.site, elem & ~(uint64_t)7, (uint32_t)p.t) >> (elem & 4);   // low or high half
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = ((keep >> e) & 1u) ? a.site.scale : 0.f;
    } else if (a.site.kind == BMI_SITE_MASKSEMBLE) {
        const float4 k4 = *(const float4*)(p.mrow + c4);
        m[0] = k4.x; m[1] = k4.y; m[2] = k4.z; m[3] = k4.w;
    }
}

template <bool BF>
__device__ __forceinline__ void epilogue_quad(const ConvArgs& a, const PixelCtx& p, float v[4], int c4) {
    if (a.scale) {
        const float4 s4 = *(const float4*)(a.scale + c4);
        v[0] *= s4.x * a.out_mul; v[1] *= s4.y * a.out_mul; v[2] *= s4.z * a.out_mul; v[3] *= s4.w * a.out_mul;
    } else if (a.out_mul != 1.f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= a.out_mul;
    }
    if (a.bias) {
        const float4 b4 = *(const float4*)(a.bias + c4);
        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
    }
    float m[4];
    site_mult4(a, p, c4, m);
    if (a.site_inner) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = m[e] == 0.f ? 0.f : v[e] * m[e];
        if (a.bias_post) {
            const float4 b4 = *(const float4*)(a.bias_post + c4);
            v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
        }
    }
    if (p.resp) {
        const half4 r4 = *(const half4*)(p.resp + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += a16_to_f32<BF>(r4[e]);
    }
    if (a.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (!a.site_inner) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = m[e] == 0.f ? 0.f : v[e] * m[e];
    }
    half4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = a16_from_f32<BF>(v[e]);
    *(half4*)(a.out + p.out_off + c4) = o;
}


// The same for the engines that keep fp32 activations (the exact engine's conv_exact.hip, the split engines' conv_split.hip):
// arithmetic and order of epilogue_quad, fp32 residual row (`resp`, or null), the finished quad left in v[] for the caller's store.
// (rq: the four residual VALUES of the quad, added when has_res — a flag, not a null pointer: a select between a local array's address
//  and null sends the array to scratch)
// (sc / bi: the folded-BN vectors indexed by c4 — a.scale / a.bias, or the second conv's of a pair launch moved back by the launch's split)
__device__ __forceinline__ void epilogue_quad_f32sb(const ConvArgs& a, const float* sc, const float* bi, const PixelCtx& p, const float* rq, bool has_res,
                                                    float v[4], int c4) {
    if (sc) {
        const float4 s4 = *(const float4*)(sc + c4);
        v[0] *= s4.x * a.out_mul; v[1] *= s4.y * a.out_mul; v[2] *= s4.z * a.out_mul; v[3] *= s4.w * a.out_mul;
    } else if (a.out_mul != 1.f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= a.out_mul;
    }
    if (bi) {
        const float4 b4 = *(const float4*)(bi + c4);
        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
    }
    float m[4];
    site_mult4(a, p, c4, m);
    if (a.site_inner) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = m[e] == 0.f ? 0.f : v[e] * m[e];
        if (a.bias_post) {
            const float4 b4 = *(const float4*)(a.bias_post + c4);
            v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
        }
    }
    if (has_res) { v[0] += rq[0]; v[1] += rq[1]; v[2] += rq[2]; v[3] += rq[3]; }
    if (a.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (!a.site_inner) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = m[e] == 0.f ? 0.f : v[e] * m[e];
    }
}
__device__ __forceinline__ void epilogue_quad_f32v(const ConvArgs& a, const PixelCtx& p, const float* rq, bool has_res, float v[4], int c4) {
    epilogue_quad_f32sb(a, a.scale, a.bias, p, rq, has_res, v, c4);
}
__device__ __forceinline__ void epilogue_quad_f32(const ConvArgs& a, const PixelCtx& p, const float* resp, float v[4], int c4) {
    float rq[4] = {0.f, 0.f, 0.f, 0.f};
    if (resp) { const float4 r4 = *(const float4*)(resp + c4); rq[0] = r4.x; rq[1] = r4.y; rq[2] = r4.z; rq[3] = r4.w; }
    epilogue_quad_f32v(a, p, rq, resp != nullptr, v, c4);
}


// ---------------------------------------------------------------------------------------------
// Coalesced epilogue for 128-channel tiles: accumulators -> LDS (fp32, after BN scale/bias) ->
// full 256-byte NHWC row segments.  The per-quad path above issues, per lane, dozens of
// dependent 8-byte loads/stores that touch 32 different rows per instruction; here every global
// access is 16 B per lane with 16 consecutive lanes covering one pixel's 128 channels, the
// residual loads of a round are all issued before the first use, and the BN vectors are loaded
// once per channel quad instead of once per (quad, pixel tile).
//   LDS tile  [128 pixels][128 ch] fp32, 512-byte rows; 16-byte chunk q of pixel row p is stored
//   at chunk q ^ (p & 31): conflict-free for the ds_write_b128 of phase 1 (8-lane groups = 8
//   different pixels, same q) and for the ds_read_b128 of phase 2.
//   A round handles 2 of the wave's TJ pixel tiles (64 KB); TJ = 4 takes two rounds.
typedef float f32x16_e __attribute__((ext_vector_type(16)));
typedef float f32x4_e __attribute__((ext_vector_type(4)));
typedef _Float16 half8_e __attribute__((ext_vector_type(8)));

#define BMI_EPILOGUE_LDS_BYTES 65536
// Timing probes (tools/ab_build.py name:-DBMI_ABL_...=1; wrong results by construction, never in the product build)
#ifndef BMI_ABL_NORES
#define BMI_ABL_NORES 0      // the lite epilogue neither fetches nor adds the residual
#endif
#ifndef BMI_EPI_NT_RES
#define BMI_EPI_NT_RES 0     // 1: epilogue_lite's residual DMA carries the non-temporal hint (A/B)
#endif
#ifndef BMI_EPI_NT_STORE
#define BMI_EPI_NT_STORE 0   // 1: the coalesced epilogues' stores carry the non-temporal hint (A/B)
#endif
#ifndef BMI_ABL_NOSTORE
#define BMI_ABL_NOSTORE 0    // the plain / lite epilogues store nothing
#endif
#ifndef BMI_DEFAULT_MFMA_SHAPE
#define BMI_DEFAULT_MFMA_SHAPE 16   // v_mfma_f32_16x16x32_f16 (same-process A/B on the headline step: 26.04 vs 26.82 ms with 32x32x16, bit-identical results)
#endif

// launch-uniform: BN + ReLU only (an inner site also has kind != NONE)
__host__ __device__ inline bool conv_epilogue_is_plain(const ConvArgs& a) { return !a.res && a.site.kind == BMI_SITE_NONE; }

// LDS-only barrier: the epilogue's global stores / outstanding residual loads must NOT be drained
// at the round boundaries (a __syncthreads() would add s_waitcnt vmcnt(0)).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Plain epilogue (no residual, no stochastic site: 15 of the 18 suffix convs of ResNet-18 multi-exit): BN and
// ReLU are applied on the accumulators, the fp16 results make ONE trip through LDS (the whole 128-channel x
// 64*TJ-pixel tile is 16*TJ KB as fp16, a single round: one barrier instead of four, half the LDS bytes) and
// leave as 256-byte NHWC row segments.  Same arithmetic as epilogue_coalesced (fma-free scale, bias, max), so a
// conv gives the same bits whichever epilogue finishes it.
//   LDS tile [64*TJ pixels][128 ch] fp16, 256-byte rows; the 16-byte chunk q (8 channels) of pixel row p lives at
//   chunk q ^ (p & 15), and its two 8-byte quads are swapped when (p >> 4) & 1: the 32 lanes of a ds_write_b64
//   (32 pixels, one channel quad) then hit 32 distinct 8-byte bank pairs.
// MS = 16 (v_mfma_f32_16x16x32 accumulators: lane = pixel l & 15 of a 16-pixel tile, registers = 4 consecutive channels
// 16*i + 4*(l >> 4) ..): the same LDS image is written from the other register layout — a lane's quad is channel quad
// 4*i + (l >> 4) of the wave's 16, i.e. 16-byte chunk 2*i + (l >> 5), half (l >> 4) & 1; 32 lanes of a ds_write_b64 are
// 16 pixels x 2 halves of one chunk column: 32 distinct 8-byte slots.
template <int TJ, int MS, bool BF, class ACC, class OffMap>
__device__ __forceinline__ void epilogue_plain(const ConvArgs& a, ACC& acc, char* lds, int tid, int ch0, OffMap offmap) {
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int wc = wave >> 1, wp = wave & 1;
    if constexpr (MS == 16) {
        const int l16 = lane & 15, q4 = lane >> 4;
        lds_barrier();   // the main loop is done with the LDS
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c4 = ch0 + wc * 64 + 16 * i + 4 * q4;
            f32x4_e sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
            if (a.scale) sc = *(const f32x4_e*)(a.scale + c4);
            if (a.bias) bi = *(const f32x4_e*)(a.bias + c4);
            sc *= a.out_mul;
            const int cq = wc * 8 + 2 * i + (q4 >> 1);
#pragma unroll
            for (int j = 0; j < 2 * TJ; ++j) {
                const int p = wp * (32 * TJ) + 16 * j + l16;
                half4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[i][j][e] * sc[e] + bi[e];
                    if (a.relu) v = fmaxf(v, 0.f);
                    o[e] = a16_from_f32<BF>(v);
                }
                *(half4*)(lds + p * 256 + ((cq ^ l16) << 4) + (((q4 ^ j) & 1) << 3)) = o;   // (p >> 4) & 1 == j & 1: quads swapped
            }
        }
    } else {
    // BN vectors of the 8 channel quads this lane holds (channel = ch0 + wc*64 + 32*i + 8*g4 + 4*hh), fetched one
    // quad ahead of their use: all eight at once would be 64 VGPRs on top of the 128 accumulators
    const int cl = ch0 + wc * 64 + 4 * hh;
#define BMI_EPI_BN(Q, SC, BI)                                                          \
    {                                                                                  \
        const int c4_ = cl + 32 * ((Q) >> 2) + 8 * ((Q) & 3);                          \
        SC = f32x4_e{1.f, 1.f, 1.f, 1.f};                                              \
        BI = f32x4_e{0.f, 0.f, 0.f, 0.f};                                              \
        if (a.scale) SC = *(const f32x4_e*)(a.scale + c4_);                            \
        if (a.bias) BI = *(const f32x4_e*)(a.bias + c4_);                              \
        SC *= a.out_mul;                                                               \
    }
    f32x4_e scn, bin;
    BMI_EPI_BN(0, scn, bin);
    const int hsw = hh ^ ((r >> 4) & 1);
    lds_barrier();   // the main loop is done with the LDS
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const f32x4_e sc = scn, bi = bin;
        if (q < 7) { BMI_EPI_BN(q + 1, scn, bin); }
        const int i = q >> 2, g4 = q & 3;
        const int cq = wc * 8 + 4 * i + g4;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int p = wp * (32 * TJ) + 32 * j + r;
            half4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[i][j][4 * g4 + e] * sc[e] + bi[e];
                if (a.relu) v = fmaxf(v, 0.f);
                o[e] = a16_from_f32<BF>(v);
            }
            *(half4*)(lds + p * 256 + ((cq ^ (r & 15)) << 4) + hsw * 8) = o;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef BMI_EPI_BN
    }
    lds_barrier();
    // phase 2: 64*TJ pixels x 16 chunks of 8 channels, 4*TJ per thread.  All LDS reads are issued before the first
    // store (the accumulators are dead: registers are free), rows beyond the tensor are only skipped at the store.
    const int k = tid & 15;
    half8_e o[4 * TJ];
#pragma unroll
    for (int it = 0; it < 4 * TJ; ++it) {
        const int pl = (tid >> 4) + 16 * it;
        o[it] = *(const half8_e*)(lds + pl * 256 + ((k ^ (pl & 15)) << 4));
    }
#pragma unroll
    for (int it = 0; it < 4 * TJ; ++it) {
        const int pl = (tid >> 4) + 16 * it;
        size_t off;
        if (!offmap(pl, off) || BMI_ABL_NOSTORE) continue;
        half8_e v = o[it];
        if (it & 1) v = __builtin_shufflevector(v, v, 4, 5, 6, 7, 0, 1, 2, 3);   // (pl >> 4) & 1 == it & 1: quads swapped
        if constexpr (BMI_EPI_NT_STORE) __builtin_nontemporal_store(v, (half8_e*)(a.out + off + ch0 + 8 * k));
        else *(half8_e*)(a.out + off + ch0 + 8 * k) = v;
    }
}

// Multipliers of the stochastic site for 8 consecutive channels c8.. of one pixel (one Philox call): 0 for a dropped
// element, 1/(1-p) for a kept one, the mask value for Masksembles, 1 without a site.  The site code is expanded ONCE per
// item and applied through `m` wherever the site sits (inner / outer): a second expansion pushed the item loop past
// hipcc's full-unroll budget, the accumulator and residual arrays became dynamically indexed and moved to scratch
// (1 KB per lane, the kernel ran 10x slower) — tests/test_build_resources.py now checks ScratchSize == 0.
__device__ __forceinline__ void site_mult8(const ConvArgs& a, const PixelCtx& px, int c8, float m[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = 1.f;
    if (a.site.kind == BMI_SITE_ELEMENTWISE || a.site.kind == BMI_SITE_CHANNEL) {
        const int row = a.site.kind == BMI_SITE_ELEMENTWISE ? px.e_pix : px.b;
        const uint64_t elem = (uint64_t)row * a.Cout + c8;
        const uint32_t keep = site_keep8(a.site, elem, (uint32_t)px.t);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = ((keep >> e) & 1u) ? a.site.scale : 0.f;
    } else if (a.site.kind == BMI_SITE_MASKSEMBLE) {
        const f32x4_e k0 = *(const f32x4_e*)(px.mrow + c8), k1 = *(const f32x4_e*)(px.mrow + c8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { m[e] = k0[e]; m[4 + e] = k1[e]; }
    }
}

// Which epilogue finishes a launch (launch-uniform; a KERNEL TEMPLATE PARAMETER, see below).
#define BMI_EPI_GENERAL 0   // anything: two fp32 rounds through LDS (epilogue_coalesced)
#define BMI_EPI_PLAIN 1     // BN + ReLU (epilogue_plain)
#define BMI_EPI_LITE 2      // BN + residual + ReLU + elementwise 2-bit or Masksembles site, one fp16 trip through LDS (epilogue_lite)
// The lite epilogue with its launch-uniform terms as COMPILE-TIME constants (same arithmetic, same bits).  As run-time values hipcc
// if-converts them inside the 64-128 times unrolled element loop: every element of a launch without a site still executes the
// compare / select / multiply of both site kinds (2 + 3 vector instructions), and a residual row costs two integer divisions
// (pixel -> image, image % res_mod).  On conv1x1_stream's Bottleneck tails that is most of the instruction stream: 3 400 instructions
// per 128 x 128 tile against 64 MFMAs, twelve waves per CU issuing 36 % of their cycles each — the kernel was issue-bound, not HBM-bound.
#define BMI_EPI_LITE_RES 3      // residual with one row per output row (res_mod >= N), ReLU, no site
#define BMI_EPI_LITE_RES_MC 4   // ... and the elementwise 2-bit MC-dropout site
#define BMI_EPI_LITE_RES_MSK 5  // ... or the Masksembles2D site (conv3x3_pw / conv3x3_patch: BASELINE configs[3])
__host__ __device__ inline int conv_epilogue_kind(const ConvArgs& a, int mfma_shape) {
    if (conv_epilogue_is_plain(a)) return BMI_EPI_PLAIN;
    const bool site_ok = a.site.kind == BMI_SITE_NONE || a.site.kind == BMI_SITE_MASKSEMBLE ||
                         (a.site.kind == BMI_SITE_ELEMENTWISE && a.site.log2_bits == 1 && !a.site.drop_all);
    if (mfma_shape == 16 && site_ok && !a.site_inner && !a.bias_post) return BMI_EPI_LITE;
    return BMI_EPI_GENERAL;
}
// ... for the kernels that instantiate the specialised forms (the others keep BMI_EPI_LITE)
__host__ __device__ inline int conv_epilogue_kind_fine(const ConvArgs& a, int mfma_shape) {
    const int k = conv_epilogue_kind(a, mfma_shape);
    if (k != BMI_EPI_LITE || !a.res || a.res_mod < a.N || a.pool || !a.relu) return k;
    if (a.site.kind == BMI_SITE_NONE) return BMI_EPI_LITE_RES;
    if (a.site.kind == BMI_SITE_ELEMENTWISE) return BMI_EPI_LITE_RES_MC;
    if (a.site.kind == BMI_SITE_MASKSEMBLE) return BMI_EPI_LITE_RES_MSK;
    return k;
}
inline int conv_epilogue_kind_launch(const ConvArgs& a, int mfma_shape) {   // host: "epilogue_lite" = 2 keeps the unspecialised form (A/B)
    return opt_epilogue_lite() == 2 ? conv_epilogue_kind(a, mfma_shape) : conv_epilogue_kind_fine(a, mfma_shape);
}
__host__ __device__ constexpr bool conv_epilogue_is_lite(int epi) {
    return epi == BMI_EPI_LITE || epi == BMI_EPI_LITE_RES || epi == BMI_EPI_LITE_RES_MC || epi == BMI_EPI_LITE_RES_MSK;
}

// "Lite" general epilogue (16x16x32 accumulators): the common non-plain launch of the path is a BasicBlock tail — BN,
// residual add, ReLU, and (p = 0.25 MC-dropout) an elementwise site drawn at 2 bits per element.  epilogue_coalesced
// handles it by moving the raw fp32 accumulators through LDS in two 64 KB rounds (four barriers, 2 x the LDS bytes, BN
// and Philox in the second phase); measured on the 16x16-map conv of the headline: 2.17 ms against 1.71 ms with the
// plain epilogue, and either feature alone already costs +15-20 %, i.e. the structure is the cost, not the arithmetic.
// Here everything happens on the accumulator registers and the tile makes ONE fp16 trip through LDS like the plain one:
//   1. BN scale/bias on the registers;
//   2. the residual tile (64*TJ pixels x 128 channels, fp16) is DMA'd into LDS in the layout of the plain epilogue's
//      output image (chunk q of pixel row p at q ^ (p & 15), no quad swap), coalesced 16-byte pieces; while it is in
//      flight each lane computes its Philox calls: one call masks the 64 channels of (pixel, wave channel half), the
//      wave's 16*2TJ pixels need 2TJ*16 calls = TJ/2 per lane, exchanged with ds_bpermute;
//   3. a lane reads the residual quad that sits in ITS output slot, adds, applies ReLU and the keep bits (word i of the
//      call, byte lane >> 4: the 4 two-bit fields of channels 16*i + 4*(lane >> 4) ..), rounds once and writes the result
//      back IN PLACE;
//   4. barrier, then the plain epilogue's store phase.
// Same arithmetic in the same order as epilogue_coalesced (fp32 throughout, one rounding), so the bits agree.
// RES_IN_LDS: the caller has issued the residual DMA of step 2 itself (same pieces, same layout, into `lds`) before or
// during its main loop — conv1x1_stream, whose tile has nothing else to overlap the residual's HBM round trip with.
// POOL (conv3x3_pw on 4x4 maps, ConvArgs::pool): the 16 pixels of an accumulator tile are one image; instead of the fp16 map the
// launch stores fp32 means over the map, [row][Cout] — the tile's 16 lanes are a DPP row: four v_add_f32 with DPP modifiers leave the
// sum in every lane, lane 0 stores 4 consecutive channels.  The conv feeds nothing but an exit head (relu -> avg_pool2d(4) -> Linear).
// SK: the site kind as a compile-time constant (BMI_SITE_NONE | BMI_SITE_ELEMENTWISE | BMI_SITE_MASKSEMBLE), or -1 = whatever the launch carries.  SK >= 0
// also says the launch HAS a residual whose rows are the output's rows (BMI_EPI_LITE_RES / _RES_MC): its address comes from `offmap`.
// RREG (conv1x1_stream's specialised forms): the caller has loaded the residual quads of this lane's accumulator positions into `rreg`
// (half4 [4][2 TJ], [i][j] = channels ch0 + wc 64 + 16 i + 4 (lane >> 4) .. of pixel wp 32 TJ + 16 j + (lane & 15)) at the START of its tile:
// no residual DMA, no wait for it and one barrier less here.
struct NoResRegs {};
template <int TJ, bool BF, bool RES_IN_LDS = false, bool POOL = false, int SK = -1, int RES_VMCNT = 0, bool NT_OUT = false, class ACC, class PixMap, class OffMap, class RREG = NoResRegs>
__device__ __forceinline__ void epilogue_lite(const ConvArgs& a, ACC& acc, char* lds, int tid, int ch0,
                                              PixMap pixmap, OffMap offmap, RREG* rreg = nullptr) {
    constexpr bool RES_REGS = !__is_same(RREG, NoResRegs);
    const int lane = tid & 63, wave = tid >> 6;
    const int wc = wave >> 1, wp = wave & 1;
    const int l16 = lane & 15, q4 = lane >> 4;
    const int HoWo = a.Ho * a.Wo;
    const bool masked = SK >= 0 ? SK == BMI_SITE_ELEMENTWISE : a.site.kind == BMI_SITE_ELEMENTWISE;
    const bool msk = SK >= 0 ? SK == BMI_SITE_MASKSEMBLE : a.site.kind == BMI_SITE_MASKSEMBLE;      // Masksembles2D: per-channel multipliers of mask (cnt0 + t) mod M
    const bool has_res = BMI_ABL_NORES ? false : (SK >= 0 ? true : a.res != nullptr);
    const bool relu = SK >= 0 ? true : a.relu != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c4 = ch0 + wc * 64 + 16 * i + 4 * q4;
        f32x4_e sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
        if (a.scale) sc = *(const f32x4_e*)(a.scale + c4);
        if (a.bias) bi = *(const f32x4_e*)(a.bias + c4);
        sc *= a.out_mul;
#pragma unroll
        for (int j = 0; j < 2 * TJ; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = acc[i][j][e] * sc[e] + bi[e];
    }
    lds_barrier();   // the main loop is done with the LDS
    if (has_res && !RES_IN_LDS && !RES_REGS) {
#pragma unroll
        for (int i = 0; i < 4 * TJ; ++i) {
            const int q = i * 256 + tid, p = q >> 4, pos = q & 15;
            const _Float16* src;
            if constexpr (SK >= 0) {
                size_t off;
                src = offmap(p, off) ? a.res + off + ch0 + ((pos ^ (p & 15)) << 3) : a.res;
            } else {
                int n, rem;
                const bool ok = pixmap(p, n, rem);
                src = ok ? a.res + ((size_t)(n % a.res_mod) * HoWo + rem) * a.Cout + ch0 + ((pos ^ (p & 15)) << 3)
                         : a.res;   // rows beyond the tensor are never stored
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lds + (i * 256 + wave * 64) * 16), 16, 0, BMI_EPI_NT_RES ? 2 : 0);
        }
    }
    philox4 mine[TJ / 2];
    if (masked) {
#pragma unroll
        for (int r = 0; r < TJ / 2; ++r) {
            int n, rem;
            pixmap(wp * (32 * TJ) + 16 * (q4 + 4 * r) + l16, n, rem);
            const int tl = n / a.B;
            const uint64_t e0 = (uint64_t)((n - tl * a.B) * HoWo + rem) * a.Cout + ch0 + wc * 64;
            mine[r] = philox_site_call(a.site, e0, (uint32_t)(a.t0 + tl));
        }
    }
    if constexpr (!RES_REGS) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RES_VMCNT) : "memory");     // (RES_VMCNT: pieces the caller issued AFTER the residual's and does not need yet)
        lds_barrier();
    }
#pragma unroll
    for (int jb = 0; jb < 2 * TJ; jb += 4) {
        half4 r4[4][4];
        if constexpr (RES_REGS) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int i = 0; i < 4; ++i) r4[jj][i] = (*rreg)[i][jb + jj];
        } else if (has_res) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int p = wp * (32 * TJ) + 16 * (jb + jj) + l16, cq = wc * 8 + 2 * i + (q4 >> 1);
                    r4[jj][i] = *(const half4*)(lds + p * 256 + ((cq ^ l16) << 4) + ((q4 & 1) << 3));
                }
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = jb + jj;
            uint32_t w[4] = {0u, 0u, 0u, 0u};
            if (masked) {   // wave-uniform
                const int src = l16 + 16 * (j & 3);
#pragma unroll
                for (int wd = 0; wd < 4; ++wd) w[wd] = (uint32_t)__shfl((int)mine[j >> 2].w[wd], src, 64);
            }
            const float* mrow = nullptr;
            if (msk) {
                int n, rem;
                pixmap(wp * (32 * TJ) + 16 * j + l16, n, rem);
                mrow = a.site.masks + (size_t)((a.site.cnt0 + a.t0 + n / a.B) % a.site.num_masks) * a.Cout + ch0 + wc * 64 + 4 * q4;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = wp * (32 * TJ) + 16 * j + l16, cq = wc * 8 + 2 * i + (q4 >> 1);
                const uint32_t fields = w[i] >> (8 * q4);
                f32x4_e mk = {1.f, 1.f, 1.f, 1.f};
                if (msk) mk = *(const f32x4_e*)(mrow + 16 * i);
                half4 o;
                f32x4_e pv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[i][j][e];
                    if (has_res) v += a16_to_f32<BF>(r4[jj][i][e]);
                    if (relu) v = fmaxf(v, 0.f);
                    if (masked) v = ((fields >> (2 * e)) & 3u) >= a.site.thresh ? v * a.site.scale : 0.f;
                    if (msk) v = mk[e] == 0.f ? 0.f : v * mk[e];
                    asm("" : "+v"(v));   // keep the fp32 product: fused into v_fma_mixlo_f16 it is rounded once instead of twice,
                                         // and 1 element in 10^6 ends one fp16 ulp away from what epilogue_coalesced stores
                    if constexpr (POOL) {
                        float x = fmaxf(v, 0.f);                       // (the head's ReLU)
                        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));    // lane ^ 1
                        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));    // lane ^ 2
                        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));   // 7 - lane (half row)
                        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));   // 15 - lane (row)
                        pv[e] = x * (1.f / 16.f);
                    } else {
                        o[e] = a16_from_f32<BF>(v);
                    }
                }
                if constexpr (POOL) {
                    int n, rem;
                    const bool okp = pixmap(p, n, rem);
                    if (l16 == 0 && okp) *(f32x4_e*)(a.pool + (size_t)n * a.Cout + ch0 + wc * 64 + 16 * i + 4 * q4) = pv;
                } else {
                    *(half4*)(lds + p * 256 + ((cq ^ l16) << 4) + ((q4 & 1) << 3)) = o;
                }
            }
        }
    }
    if constexpr (POOL) return;
    lds_barrier();
    const int k = tid & 15;
    half8_e o[4 * TJ];
#pragma unroll
    for (int it = 0; it < 4 * TJ; ++it) {
        const int pl = (tid >> 4) + 16 * it;
        o[it] = *(const half8_e*)(lds + pl * 256 + ((k ^ (pl & 15)) << 4));
    }
#pragma unroll
    for (int it = 0; it < 4 * TJ; ++it) {
        const int pl = (tid >> 4) + 16 * it;
        size_t off;
        if (!offmap(pl, off) || BMI_ABL_NOSTORE) continue;
        if constexpr (NT_OUT || BMI_EPI_NT_STORE) __builtin_nontemporal_store(o[it], (half8_e*)(a.out + off + ch0 + 8 * k));     // (conv1x1_seam: a streamed tensor must not evict the tile's re-read operands from L2)
        else *(half8_e*)(a.out + off + ch0 + 8 * k) = o[it];
    }
}

// pixmap(p, n, rem) -> bool: tile pixel p -> image n and y*Wo+x (false beyond the tensor);
// offmap(p, off) -> bool: the same pixel's element offset in the output tensor (no division for linear tiles).
// EPI (chosen per launch by conv_epilogue_kind) selects the epilogue at compile time: with two paths in one kernel the
// register allocator spills 56-80 VGPRs in the other one.
template <int TJ, int EPI, int MS, bool BF, class ACC, class PixMap, class OffMap>
__device__ __forceinline__ void epilogue_coalesced(const ConvArgs& a, ACC& acc, char* lds, int tid, int ch0,
                                                   PixMap pixmap, OffMap offmap) {
    static_assert(TJ == 2 || TJ == 4, "two pixel tiles per round");
    static_assert(MS == 32 || MS == 16, "MFMA shape");
    if constexpr (EPI == BMI_EPI_PLAIN) {
        epilogue_plain<TJ, MS, BF>(a, acc, lds, tid, ch0, offmap);
        return;
    }
    if constexpr (conv_epilogue_is_lite(EPI)) {
        static_assert(MS == 16, "the lite epilogue reads the 16x16x32 accumulator layout");
        constexpr int SK = EPI == BMI_EPI_LITE_RES ? BMI_SITE_NONE : (EPI == BMI_EPI_LITE_RES_MC ? BMI_SITE_ELEMENTWISE : (EPI == BMI_EPI_LITE_RES_MSK ? BMI_SITE_MASKSEMBLE : -1));
        epilogue_lite<TJ, BF, false, false, SK>(a, acc, lds, tid, ch0, pixmap, offmap);
        return;
    }
    constexpr int NR = TJ / 2;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int wc = wave >> 1, wp = wave & 1;
    const int k = tid & 15;
    // Residual rows are requested one round ahead (round 0 here, round r+1 right after the LDS
    // writes of round r): their HBM latency overlaps the BN pass / LDS transposition instead of
    // being exposed in every round, and at most two rounds' rows (64 VGPRs) are live at once.
    half8_e resv[NR][8];
    int pn[NR][8], prem[NR][8];   // image index (-1: tile pixel beyond the tensor) and y*Wo+x
#define BMI_EPI_FETCH(RR)                                                                              \
    _Pragma("unroll") for (int it = 0; it < 8; ++it) {                                                 \
        const int pl_ = (tid >> 4) + 16 * it;                                                          \
        const int p_ = (pl_ >> 6) * (32 * TJ) + (RR) * 64 + (pl_ & 63); /* pixel inside the WG tile */ \
        int n_, rem_;                                                                                  \
        const bool okp_ = pixmap(p_, n_, rem_);                                                        \
        pn[RR][it] = okp_ ? n_ : -1;                                                                   \
        prem[RR][it] = rem_;                                                                           \
        if (okp_ && a.res)                                                                             \
            resv[RR][it] = *(const half8_e*)(a.res + ((size_t)(n_ % a.res_mod) * (a.Ho * a.Wo) + rem_) * a.Cout + ch0 + 8 * k); \
    }
    BMI_EPI_FETCH(0);
    // folded-BN scale / bias of the 8 channels this thread finishes in phase 2
    f32x4_e sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, bi0 = {0.f, 0.f, 0.f, 0.f}, bi1 = bi0;
    if (a.scale) { sc0 = *(const f32x4_e*)(a.scale + ch0 + 8 * k); sc1 = *(const f32x4_e*)(a.scale + ch0 + 8 * k + 4); }
    if (a.bias) { bi0 = *(const f32x4_e*)(a.bias + ch0 + 8 * k); bi1 = *(const f32x4_e*)(a.bias + ch0 + 8 * k + 4); }
    sc0 *= a.out_mul;
    sc1 *= a.out_mul;
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        lds_barrier();   // main loop (or previous round's phase 2) is done with the LDS
        // phase 1: raw fp32 accumulators into the swizzled LDS tile (BN is applied in phase 2, where
        // a thread owns the same 8 channels for all its rows, so the BN vectors are loaded once)
        if constexpr (MS == 16) {
            // 16x16x32 accumulators: a register quad IS 4 consecutive channels (16*i + 4*(lane >> 4) ..) of pixel lane & 15;
            // 16 consecutive lanes = 16 pixels, one chunk column: 16 distinct 16-byte slots per ds_write_b128 group
            const int l16 = lane & 15, q4 = lane >> 4;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int pl = wp * 64 + jj * 16 + l16;   // pixel inside the round
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4_e v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][4 * rr + jj][e];
                    const int cq = wc * 16 + 4 * i + q4;
                    *(f32x4_e*)(lds + pl * 512 + ((cq ^ (pl & 31)) << 4)) = v;
                }
            }
        } else {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int pl = wp * 64 + jj * 32 + r;   // pixel inside the round
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = q >> 2, g4 = q & 3;
                f32x4_e v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][2 * rr + jj][4 * g4 + e];
                const int cq = wc * 16 + 8 * i + 2 * g4 + hh;
                *(f32x4_e*)(lds + pl * 512 + ((cq ^ (pl & 31)) << 4)) = v;
            }
        }
        }
        if (rr + 1 < NR) { BMI_EPI_FETCH(rr + 1); }
        lds_barrier();
        // phase 2: 128 pixels x 16 groups of 8 channels = 2048 items, 8 per thread
        // Shared Philox for elementwise sites at 2 bits per element (p = 0.25, 0.5, 0.75): one call masks 64 channels
        // of a pixel, i.e. the items of 8 neighbouring lanes.  A 16-lane group needs 8 pixels x 2 halves = 16 calls in
        // this round: lane k computes the one of pixel (k & 7), half (k >> 3), and every item fetches its call's words
        // from lane (k & 8) | it with ds_bpermute: ONE Philox per thread and round instead of eight.
#ifndef BMI_EPI_SHARE
#define BMI_EPI_SHARE 1
#endif
        const bool share = BMI_EPI_SHARE && a.site.kind == BMI_SITE_ELEMENTWISE && a.site.log2_bits == 1 && !a.site.drop_all;
        philox4 mine = {{0u, 0u, 0u, 0u}};
        if (share) {
            const int pl_ = (tid >> 4) + 16 * (k & 7);
            int n_, rem_;
            pixmap((pl_ >> 6) * (32 * TJ) + rr * 64 + (pl_ & 63), n_, rem_);
            const int tl_ = n_ / a.B;
            const uint64_t e0_ = (uint64_t)((n_ - tl_ * a.B) * (a.Ho * a.Wo) + rem_) * a.Cout + ch0 + 64 * (k >> 3);
            mine = philox_site_call(a.site, e0_, (uint32_t)(a.t0 + tl_));
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            philox4 theirs = mine;
            if (share) {   // wave-uniform: every lane takes part in the exchange, valid pixel or not
                const int src = (lane & 48) | (k & 8) | it;
#pragma unroll
                for (int wd = 0; wd < 4; ++wd) theirs.w[wd] = (uint32_t)__shfl((int)mine.w[wd], src, 64);
            }
            if (pn[rr][it] < 0) continue;
            const int pl = (tid >> 4) + 16 * it;
            const int s = pl & 31;
            const PixelCtx px = make_pixel_ctx(a, pn[rr][it], prem[rr][it]);
            const f32x4_e lo = *(const f32x4_e*)(lds + pl * 512 + (((2 * k) ^ s) << 4));
            const f32x4_e hi = *(const f32x4_e*)(lds + pl * 512 + (((2 * k + 1) ^ s) << 4));
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = lo[e] * sc0[e] + bi0[e]; v[4 + e] = hi[e] * sc1[e] + bi1[e]; }
            const int c8 = ch0 + 8 * k;
            float m[8];
            if (share) {
                const uint32_t keep = philox_keep8(theirs, (uint32_t)((uint64_t)px.e_pix * a.Cout + c8), 1, a.site.thresh);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = ((keep >> e) & 1u) ? a.site.scale : 0.f;
            } else {
                site_mult8(a, px, c8, m);
            }
            if (a.site_inner) {   // mask between the conv and its BatchNorm shift (converter/pytorch rule)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = m[e] == 0.f ? 0.f : v[e] * m[e];
                if (a.bias_post) {
                    const f32x4_e p0 = *(const f32x4_e*)(a.bias_post + c8), p1 = *(const f32x4_e*)(a.bias_post + c8 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += p0[e]; v[4 + e] += p1[e]; }
                }
            }
            if (px.resp) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += a16_to_f32<BF>(resv[rr][it][e]);
            }
            if (a.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (!a.site_inner) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = m[e] == 0.f ? 0.f : v[e] * m[e];
            }
            half8_e o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = a16_from_f32<BF>(v[e]);
            if constexpr (BMI_EPI_NT_STORE) __builtin_nontemporal_store(o, (half8_e*)(a.out + px.out_off + c8));
            else *(half8_e*)(a.out + px.out_off + c8) = o;
        }
    }
}
#undef BMI_EPI_FETCH
