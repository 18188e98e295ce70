// HBM-bound 1x1 convolutions (the expand / reduce convs of Bottleneck nets, SA-style ResNet-50: a few hundred FLOPs per
// byte at most, K = Cin <= 512): a kernel shaped for memory overlap instead of MFMA efficiency.
//
// conv_igemm_wide gives these launches a 256 x 256 tile, 8 waves and 128-136 KB of LDS: ONE workgroup per CU, whose life is
// a chain of memory phases that never overlap (operand DMA -> wait -> 2-4 K-steps -> residual DMA -> wait -> stores).
// Measured (tools/per_launch.py, ResNet-50 multi-exit, 16000 image-samples): the 128 -> 512 tails run at 3.5-3.7 TB/s of
// algorithmic bytes, the 256 -> 1024 ones at 3.0, while a plain `out = a + b` over the same three tensors streams 6.1 TB/s
// (tools/experiments/hbm_streams.py).  Here:
//   tile        = 128 channels x 128 pixels (first version: 256), 4 waves (each 64 ch x 64 px, v_mfma_f32_16x16x32), 256 threads;
//   LDS         = 32 KB: one single-buffered [weights 16 KB | pixels 16 KB] K-step of 64 channels, then the epilogue's
//                 output image -> THREE or FOUR workgroups per CU: while one waits for its operands or its residual, the
//                 others compute or store.  No intra-workgroup pipelining at all: the overlap is between workgroups;
//   operands    = LDS-DMA (global_load_lds), 128-byte rows with the XOR swizzle of conv_igemm_wide on the source side;
//   epilogue    = conv_epilogue.h (plain / lite / general), the same code and the same bits as every other conv kernel.
// Taken for ksize 1, pad 0, stride 1 or 2, Cin % 64 == 0, Cin <= 512, Cout % 128 == 0 (conv_takes_stream_kernel) when the launch
// carries a residual (the Bottleneck tails); the engine tries it before conv_igemm_wide.  Selection depends on the conv's
// shape and epilogue terms only.  Measured: 128 -> 512 tail 2502 -> 2227 us, 256 -> 1024 tail 1529 -> 1373 us (bit-identical
// results); still 4.2 / 3.4 TB/s against the 6.1 TB/s of an elementwise kernel: each workgroup exposes three HBM round trips
// (operands per K-step, residual) and two workgroups per CU do not cover them all.
#include "conv_epilogue.h"
#include "kernels.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define GLDS16(SRC, LDSPTR)                                                                     \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC),       \
                                     (__attribute__((address_space(3))) void*)(LDSPTR), 16, 0, 0)

#define SBC 128

// TJ = 2 (default): 128-pixel tile, 32 KB of LDS, three (residual launches: 164 VGPRs) or four workgroups per CU.
// TJ = 4 ("conv_stream" = 3, the first version): 256-pixel tile, 64 KB, two workgroups per CU.  Same-process A/B at 16000
// image-samples, tails 128 -> 512 / 256 -> 1024 / 512 -> 2048: 2467 / 1476 / 1027 us with TJ = 4, 2245 / 1370 / 963 us with TJ = 2.
#ifndef STREAM_RES_PREFETCH
#define STREAM_RES_PREFETCH 0   // 1: the residual tile is DMA'd into its own 16*TJ KB right behind the first K-step's operands
#endif
#ifndef STREAM_RES_REGS
#define STREAM_RES_REGS 0       // 1: the specialised residual forms load the residual straight into registers (accumulator layout) at the START of the
                                // tile: no residual DMA, no wait for it, one barrier less.  Measured 13-17 % SLOWER (128 -> 512 tails 1.90 -> 2.15 ms,
                                // 256 -> 1024 1.14 -> 1.29, 512 -> 2048 0.76 -> 0.89): 148 VGPRs = three workgroups per CU instead of four, and 8-byte
                                // loads at a 1 KB stride; same bits (profiles/experiments/r4_stream_ablation.log)
#endif
#ifndef STREAM_ABL_NOMFMA
#define STREAM_ABL_NOMFMA 0   // timing probes (wrong results by construction): no MFMAs | no activation DMA | no keep-byte loads and masking
#endif
#ifndef STREAM_ABL_NOX
#define STREAM_ABL_NOX 0
#endif
#ifndef STREAM_ABL_NOKB
#define STREAM_ABL_NOKB 0
#endif
#ifndef STREAM_MAX_CIN
#define STREAM_MAX_CIN 512
#endif
#ifndef STREAM_KD
#define STREAM_KD 64            // channels per barrier pair: 64 | 128 (two [weights | pixels] sub-tiles per wait)
#endif
// MSK = keep bits on the input (ConvArgs::in_bits; lazy sites, engine.hip): `in` is the deterministic tensor pre-scaled by 1/(1-p), a
// thread fetches the keep byte of each pixel piece it DMAs and clears the dropped elements of ITS 16 bytes in LDS between the K-step's
// wait and its barrier (the K-steps are lock-step with a full vmcnt(0) anyway).
template <int EPI, bool BF, int TJ = 4, bool MSK = false>
__global__ __launch_bounds__(256, TJ == 4 ? 2 : ((STREAM_RES_PREFETCH && conv_epilogue_is_lite(EPI)) || STREAM_KD == 128 ? 2 : 3)) void conv1x1_stream_kernel(ConvArgs a) {
    static_assert(!MSK || (TJ == 2 && STREAM_KD == 64), "masked input: the default form only");
    constexpr int SBP = 64 * TJ;
    constexpr int NSUB = TJ == 2 ? STREAM_KD / 64 : 1;
    constexpr int SUB = (TJ == 4 ? BMI_EPILOGUE_LDS_BYTES : BMI_EPILOGUE_LDS_BYTES / 2);     // one [weights | pixels] sub-tile = the epilogue's tile
    constexpr bool RPRE = STREAM_RES_PREFETCH && conv_epilogue_is_lite(EPI) && TJ == 2 && NSUB == 1;
    __shared__ __attribute__((aligned(16))) char smem[NSUB * SUB + (RPRE ? SUB : 0)];
    constexpr int XBASE = SBC * 128;   // pixel tile behind the weight tile
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const int wc = wave >> 1, wp = wave & 1;

    int ptile, ctile;
    const int HoWo = a.Ho * a.Wo;
    if (MSK && a.lazy_order) {                        // (launcher: Ho Wo % SBP == 0, N % in_mod == 0)
        if (!lazy_tile_map(blockIdx.x, a.in_mod, a.N / a.in_mod, HoWo / SBP, a.Cout / SBC, ptile, ctile)) return;
    } else {
        xcd_tile_map(blockIdx.x, (a.M + SBP - 1) / SBP, a.Cout / SBC, ptile, ctile, a.xcd_split);
    }
    const int ch0 = ctile * SBC;
    const int pix0 = ptile * SBP;

    // DMA piece q = tid + 256*i -> tile row (q >> 3) = 32*i + (tid >> 3), 16-byte slot tid & 7 (swizzled on the source side)
    const int rowt = tid >> 3;
    const int srcchunk = ((tid & 7) ^ ((rowt >> 1) & 7)) * 8;   // (32*i >> 1) & 7 == 0: the same for all i
    const _Float16* wsrc[4];
    const _Float16* xsrc[2 * TJ];
    int bsrc[MSK ? 2 * TJ : 1];                       // MSK: byte offset of the piece's keep byte at channel chunk 0, -1 beyond the tensor
#pragma unroll
    for (int i = 0; i < 4; ++i) wsrc[i] = a.wgt + (size_t)(ch0 + 32 * i + rowt) * a.Cin + srcchunk;
#pragma unroll
    for (int i = 0; i < 4; ++i) GLDS16(wsrc[i], smem + (i * 256 + wave * 64) * 16);
    // Input row of output pixel m.  The general map costs three integer divisions per piece (pixel -> image, row, image % in_mod) — with
    // the residual rows of the epilogue, a third of this kernel's instructions.  Two launch-uniform shortcuts cover every launch of the
    // path: (1) stride 1 on a tensor with an image per output image: the row IS m; (2) a tile inside ONE image (Ho Wo % tile == 0: the
    // 32x32 / 16x16 maps): the image comes from one scalar division per workgroup, and row / column from nothing (stride 1) or a shift
    // (Wo a power of two).
    const bool lin_in = a.stride == 1 && a.in_mod >= a.N;
    const bool one_img = HoWo % SBP == 0 && (a.stride == 1 || (a.Wo & (a.Wo - 1)) == 0);
    const int n_tile = pix0 / HoWo, rem_tile = pix0 - n_tile * HoWo;     // (scalar)
    const int wo_log2 = 31 - __builtin_clz(a.Wo);
#pragma unroll
    for (int i = 0; i < 2 * TJ; ++i) {
        const int m = pix0 + 32 * i + rowt;
        const int mm = m < a.M ? m : 0;               // rows beyond the tensor read pixel 0: computed, never stored
        if (lin_in && !MSK) {
            xsrc[i] = a.in + (size_t)mm * a.Cin + srcchunk;
        } else if (one_img) {                         // (M is a multiple of Ho Wo: the whole tile lies inside the tensor)
            const int rem = rem_tile + 32 * i + rowt;
            const int pix_in = a.stride == 1 ? rem : ((rem >> wo_log2) * a.stride) * a.W + (rem & (a.Wo - 1)) * a.stride;
            xsrc[i] = a.in + ((size_t)(n_tile % a.in_mod) * a.H * a.W + pix_in) * a.Cin + srcchunk;
            if constexpr (MSK) bsrc[i] = (int)(((size_t)n_tile * a.H * a.W + pix_in) * (a.Cin >> 3) + (srcchunk >> 3));
        } else {
            const int n = mm / HoWo;
            const int rem = mm - n * HoWo;
            const int oy = rem / a.Wo;
            const int ox = rem - oy * a.Wo;
            xsrc[i] = a.in + ((size_t)(n % a.in_mod) * a.H * a.W + (size_t)(oy * a.stride) * a.W + ox * a.stride) * a.Cin + srcchunk;
            if constexpr (MSK) bsrc[i] = m < a.M ? (int)((((size_t)n * a.H + (size_t)(oy * a.stride)) * a.W + ox * a.stride) * (a.Cin >> 3) + (srcchunk >> 3)) : -1;
        }
    }
    uint32_t kb[MSK ? 2 * TJ : 1];
#define LOAD_KB(KS)                                                                               \
    if constexpr (MSK && !STREAM_ABL_NOKB) {                                                                          \
        _Pragma("unroll") for (int i = 0; i < 2 * TJ; ++i) kb[i] = bsrc[i] >= 0 ? a.in_bits[(size_t)(unsigned)bsrc[i] + (KS) * 8] : 0xffu; \
    }
#define APPLY_KB()                                                                                \
    if constexpr (MSK && !STREAM_ABL_NOKB) {                                                                          \
        typedef unsigned int u32x4_s __attribute__((ext_vector_type(4)));                         \
        _Pragma("unroll") for (int i = 0; i < 2 * TJ; ++i) {                                      \
            u32x4_s* const pp = (u32x4_s*)(smem + XBASE + (i * 256 + tid) * 16);                  \
            u32x4_s v = *pp;                                                                      \
            const int b = (int)kb[i];                                                             \
            _Pragma("unroll") for (int d = 0; d < 4; ++d) {                                       \
                const unsigned lo = (unsigned)__builtin_amdgcn_sbfe(b, 2 * d, 1), hi = (unsigned)__builtin_amdgcn_sbfe(b, 2 * d + 1, 1); \
                v[d] &= (lo & 0xffffu) | (hi & 0xffff0000u);                                      \
            }                                                                                     \
            *pp = v;                                                                              \
        }                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                        \
    }
    LOAD_KB(0);
#pragma unroll
    for (int i = 0; i < 2 * TJ; ++i) if (!STREAM_ABL_NOX) GLDS16(xsrc[i], smem + XBASE + (i * 256 + wave * 64) * 16);
    const int nK = a.Cin / 64;
#pragma unroll
    for (int u = 1; u < NSUB; ++u)
        if (u < nK) {
#pragma unroll
            for (int i = 0; i < 4; ++i) GLDS16(wsrc[i] + u * 64, smem + u * SUB + (i * 256 + wave * 64) * 16);
#pragma unroll
            for (int i = 0; i < 2 * TJ; ++i) GLDS16(xsrc[i] + u * 64, smem + u * SUB + XBASE + (i * 256 + wave * 64) * 16);
        }
    // (STREAM_RES_REGS, off) the residual quads of this lane's accumulator positions, in flight from the start of the tile
    constexpr bool RREG = STREAM_RES_REGS && !RPRE && TJ == 2 && (EPI == BMI_EPI_LITE_RES || EPI == BMI_EPI_LITE_RES_MC);
    typedef half4 rres_t[4][2 * TJ];
    rres_t rres;
    if constexpr (RREG) {
#pragma unroll
        for (int j = 0; j < 2 * TJ; ++j) {
            const int m = pix0 + wp * (32 * TJ) + 16 * j + r;
            const _Float16* rp = a.res + (size_t)(m < a.M ? m : 0) * a.Cout + ch0 + wc * 64 + 4 * kq;    // (rows beyond the tensor: row 0, never stored)
#pragma unroll
            for (int i = 0; i < 4; ++i) rres[i][j] = *(const half4*)(rp + 16 * i);
        }
    }
    if constexpr (RPRE) {
        // the residual tile, in epilogue_lite's layout (chunk q of pixel row p at q ^ (p & 15)), behind the first K-step's operands
        if (a.res) {
#pragma unroll
            for (int i = 0; i < 4 * TJ; ++i) {
                const int q = i * 256 + tid, p = q >> 4, pos = q & 15;
                const int m = pix0 + p;
                const _Float16* src;
                if constexpr (conv_epilogue_is_lite(EPI) && EPI != BMI_EPI_LITE)   // (one residual row per output row)
                    src = m < a.M ? a.res + (size_t)m * a.Cout + ch0 + ((pos ^ (p & 15)) << 3) : a.res;
                else
                    src = m < a.M ? a.res + ((size_t)((m / HoWo) % a.res_mod) * HoWo + (m % HoWo)) * a.Cout + ch0 + ((pos ^ (p & 15)) << 3) : a.res;
                GLDS16(src, smem + SUB + (i * 256 + wave * 64) * 16);
            }
        }
    }

    typedef float accv __attribute__((ext_vector_type(4)));
    accv acc[4][2 * TJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2 * TJ; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

    const int a_off = (wc * 64 + r) * 128;
    const int b_off = XBASE + (wp * (32 * TJ) + r) * 128;
    const int sw_r = (r >> 1) & 7;
    for (int ks = 0; ks < nK; ks += NSUB) {
        // (vmcnt retires in order: the prefetched residual sits BEHIND the first K-step's operands and in front of the later ones)
        if (RPRE && ks == 0 && a.res) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * TJ) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        APPLY_KB();
        __builtin_amdgcn_s_barrier();                 // K-step ks has landed (every wave waited for its own pieces)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            if (ks + u >= nK) break;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int coff = ((4 * sub + kq) ^ sw_r) << 4;
                half8 af[4], bf[2 * TJ];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *(const half8*)(smem + u * SUB + a_off + i * 16 * 128 + coff);
#pragma unroll
                for (int j = 0; j < 2 * TJ; ++j) bf[j] = *(const half8*)(smem + u * SUB + b_off + j * 16 * 128 + coff);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2 * TJ; ++j) if (!STREAM_ABL_NOMFMA || (i == 0 && j == 0)) acc[i][j] = mfma_16x16x32<BF>(af[i], bf[j], acc[i][j]);
            }
        }
        if (ks + NSUB < nK) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();             // every wave has read K-step ks: the buffer may be refilled
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < NSUB; ++u) {
                if (ks + NSUB + u >= nK) break;
                const int koff = (ks + NSUB + u) * 64;
#pragma unroll
                for (int i = 0; i < 4; ++i) GLDS16(wsrc[i] + koff, smem + u * SUB + (i * 256 + wave * 64) * 16);
#pragma unroll
                for (int i = 0; i < 2 * TJ; ++i) if (!STREAM_ABL_NOX) GLDS16(xsrc[i] + koff, smem + u * SUB + XBASE + (i * 256 + wave * 64) * 16);
            }
            LOAD_KB(ks + NSUB);
        }
    }
#undef LOAD_KB
#undef APPLY_KB

    auto pixmap = [&](int p, int& n, int& rem) -> bool {
        const int m = pix0 + p;
        if (HoWo % SBP == 0) {                        // the tile lies inside one image
            n = n_tile;
            rem = rem_tile + p;
            return true;
        }
        n = m / HoWo;
        rem = m - n * HoWo;
        return m < a.M;
    };
    auto offmap = [&](int p, size_t& off) -> bool {
        off = (size_t)(pix0 + p) * a.Cout;
        return pix0 + p < a.M;
    };
    constexpr int SK = EPI == BMI_EPI_LITE_RES ? BMI_SITE_NONE : (EPI == BMI_EPI_LITE_RES_MC ? BMI_SITE_ELEMENTWISE : -1);
    if constexpr (RREG) epilogue_lite<TJ, BF, false, false, SK>(a, acc, smem, tid, ch0, pixmap, offmap, &rres);
    else if constexpr (RPRE) epilogue_lite<TJ, BF, true, false, SK>(a, acc, smem + SUB, tid, ch0, pixmap, offmap);
    else epilogue_coalesced<TJ, EPI, 16, BF>(a, acc, smem, tid, ch0, pixmap, offmap);
}

bool conv_takes_stream_kernel(int ksize, int stride, int pad, int cin, int cout) {
    return ksize == 1 && pad == 0 && (stride == 1 || stride == 2) && cin % 64 == 0 && cin <= STREAM_MAX_CIN && cout % SBC == 0;
}

// BMI_ERR_UNSUPPORTED -> the caller goes on to conv_igemm_wide / conv_igemm.
int launch_conv1x1_stream(const ConvArgs& a_in, hipStream_t s) {
    if (!opt_conv_stream() || a_in.wgt_b || a_in.in2 || a_in.imap || (a_in.in_bits && a_in.lazy_planar)) return BMI_ERR_UNSUPPORTED;
    if (a_in.in_bits && (a_in.out_mul != 1.f || (size_t)a_in.N * a_in.H * a_in.W * (a_in.Cin >> 3) >= 0x7fffffffull)) return BMI_ERR_UNSUPPORTED;
    if (!conv_takes_stream_kernel(a_in.ksize, a_in.stride, a_in.pad, a_in.Cin, a_in.Cout)) return BMI_ERR_UNSUPPORTED;
    if (a_in.N <= 0 || a_in.M <= 0 || a_in.in_mod <= 0 || a_in.B <= 0 || (a_in.res && a_in.res_mod <= 0)) return BMI_ERR_INVALID;
    ConvArgs a = a_in;
    a.xcd_split = xcd_split_for(a.Cout / SBC, (size_t)a.Cout * a.Cin * 2);
    const int SBP = (opt_conv_stream() == 3 && !a_in.in_bits) ? 256 : 128;   // 3: the 256-pixel tile (A/B)
    const long tiles = (((long)a.M + SBP - 1) / SBP) * (a.Cout / SBC);
    if (tiles > 0x7fffffffL) return BMI_ERR_INVALID;
    // the minimum-grid rule of the other wide-tile kernels (on the engine's full-chunk image count, never this launch's)
    static const int n_cu = [] {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cu = 0;
        return cu;
    }();
    const long tiles_sel = a.n_ref > 0 ? (((long)a.n_ref * a.Ho * a.Wo + SBP - 1) / SBP) * (a.Cout / SBC) : tiles;
    if (opt_conv_stream() != 2 && tiles_sel < (n_cu > 0 ? 3 * n_cu / 2 : 384)) return BMI_ERR_UNSUPPORTED;
    int epi = opt_epilogue_lite() ? conv_epilogue_kind_launch(a, 16) : (conv_epilogue_is_plain(a) ? BMI_EPI_PLAIN : BMI_EPI_GENERAL);
    if (epi == BMI_EPI_LITE_RES_MSK) epi = BMI_EPI_LITE;      // (no Masksembles instantiation of this kernel)
    // (the general epilogue — Masksembles / channel sites, 4-16-bit probabilities — spills 16 VGPRs next to this kernel's twelve
    //  DMA row pointers: those launches stay with conv_igemm_wide)
    if (epi == BMI_EPI_GENERAL) return BMI_ERR_UNSUPPORTED;
    // Plain launches (no residual: one read stream, one write stream) stay with the persistent conv_igemm_wide, which prefetches
    // the next tile under its epilogue: same-process A/B at 16000 image-samples, 128 -> 512 / 256 -> 1024: 1333 / 914 us there,
    // 1470 / 970 us here; with the residual 2502 / 1529 us there, 2227 / 1373 us here.  "conv_stream" = 2 takes them too (tests).
    // Where the wide kernel cannot go (Cout % 256 != 0: the 256 -> 128 / 64 -> 128 reduce and downsample convs) the alternative is the
    // per-tap conv_igemm, and this kernel wins plain launches too: 256 -> 128 on 32x32 2723 -> 2310 us, 64 -> 128 stride 2 455 -> 312 us.
    if (epi == BMI_EPI_PLAIN && a.Cout % 256 == 0 && opt_conv_stream() != 2 && !a.in_bits) return BMI_ERR_UNSUPPORTED;   // (keep bits: the wide kernel has no place to apply them)
    // keep bits on a deterministic input: the samples of one activation tile back to back on one XCD (lazy_tile_map)
    a.lazy_order = a.in_bits && opt_lazy_order() && a.N % a.in_mod == 0 && (a.Ho * a.Wo) % SBP == 0 && a.in_mod < a.N;
    const dim3 grid((unsigned)(a.lazy_order ? lazy_tile_grid(tiles) : tiles)), block(256);
#define STREAM_LAUNCH(BF_)                                                                                                      \
    {                                                                                                                           \
        if (a.in_bits) {                                                                                                        \
            if (epi == BMI_EPI_PLAIN) hipLaunchKernelGGL((conv1x1_stream_kernel<BMI_EPI_PLAIN, BF_, 2, true>), grid, block, 0, s, a); \
            else hipLaunchKernelGGL((conv1x1_stream_kernel<BMI_EPI_LITE, BF_, 2, true>), grid, block, 0, s, a);                 \
        } else if (SBP == 128) {                                                                                                       \
            if (epi == BMI_EPI_PLAIN) hipLaunchKernelGGL((conv1x1_stream_kernel<BMI_EPI_PLAIN, BF_, 2>), grid, block, 0, s, a); \
            else if (epi == BMI_EPI_LITE_RES) hipLaunchKernelGGL((conv1x1_stream_kernel<BMI_EPI_LITE_RES, BF_, 2>), grid, block, 0, s, a); \
            else if (epi == BMI_EPI_LITE_RES_MC) hipLaunchKernelGGL((conv1x1_stream_kernel<BMI_EPI_LITE_RES_MC, BF_, 2>), grid, block, 0, s, a); \
            else hipLaunchKernelGGL((conv1x1_stream_kernel<BMI_EPI_LITE, BF_, 2>), grid, block, 0, s, a);                       \
        } else if (epi == BMI_EPI_PLAIN) hipLaunchKernelGGL((conv1x1_stream_kernel<BMI_EPI_PLAIN, BF_>), grid, block, 0, s, a); \
        else hipLaunchKernelGGL((conv1x1_stream_kernel<BMI_EPI_LITE, BF_>), grid, block, 0, s, a);                              \
    }
    if (a.bf16) STREAM_LAUNCH(true) else STREAM_LAUNCH(false)
#undef STREAM_LAUNCH
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}
