// Implicit-GEMM convolution for gfx950, fp16 operands / fp32 accumulate, NHWC activations.
//
//   D[cout][pixel] = sum_k W[cout][k] * X[k][pixel],   k = (ky*ks + kx)*Cin + ci
//
// Orientation: output CHANNELS ride the MFMA row axis (A operand = weights), PIXELS the column
// axis (B operand = activations).  With v_mfma_f32_32x32x16_f16 a lane then owns one pixel and,
// per accumulator quad (reg & 3), four CONSECUTIVE channels: one Philox4x32 call masks exactly
// those four values, BN scale/bias come in as float4, and the NHWC store is an 8-byte write.
// Both operands are K-contiguous per row (weights [Cout][k], activations [pixel][Cin]) so each
// fragment is one ds_read_b128.
//
// Tile: BC channels x BP pixels x 64 deep, 256 threads (4 waves, WC x WP), double-buffered LDS
// (one barrier per K-step), register-staged global loads issued before the MFMAs of the current
// step.  LDS rows are 128 B; 16-byte chunk c of row r lives at chunk (c ^ ((r >> 1) & 7)), which
// makes every ds_read_b128 lane group ({0-3,12-15,20-27}, ...) hit 16 distinct 16-B slots.
// Epilogue (fused): folded-BN scale/bias, residual add, ReLU, stochastic site (MCDropout via
// Philox, channel dropout, or Masksembles channel mask), fp16 store.
// Reference semantics: BasicBlock.forward SA/models/resnet18/resnet18.py:32-48, exit-head convs
// :306-308/:318-319/:329, MCDropout :207-210, Masksembles2D SA/utils.py:165-169.
#include <cstdlib>

#include "conv_epilogue.h"
#include "kernels.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // first-class 16-byte vector (HIP's uint4 is a struct)

#define BK 64

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int BC, int BP, int WC, int WP, bool DBUF, bool XMASK, bool PLAIN, bool BF, bool IMAP = false, bool SPLITK = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvArgs a) {
    constexpr int TI = BC / WC / 32;   // 32x32 MFMA tiles per wave along channels
    constexpr int TJ = BP / WP / 32;   // ... along pixels
    constexpr int WROWS = BC / 32;     // weight rows staged per thread
    constexpr int XROWS = BP / 32;     // pixel rows staged per thread
    constexpr int TILE = (BC + BP) * 128;
    constexpr int MAIN_BYTES = (DBUF ? 2 : 1) * TILE;
    constexpr int LDS_BYTES = (BC == 128 && MAIN_BYTES < BMI_EPILOGUE_LDS_BYTES) ? BMI_EPILOGUE_LDS_BYTES : MAIN_BYTES;
    // XMASK: keep byte -> the four dword masks of its 8 halves, a 256-entry table behind the tiles (conv3x3_s2's S2_MASK_LUT: one
    // ds_read_b128 instead of ~20 vector instructions per staged row — this kernel's masked form was issue-bound, see GLOAD)
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES + (XMASK ? 4096 : 0)];
    if constexpr (XMASK) {
        const int t = threadIdx.x;
        u32x4 m;
#pragma unroll
        for (int i = 0; i < 4; ++i) m[i] = (((t >> (2 * i)) & 1) ? 0xffffu : 0u) | (((t >> (2 * i + 1)) & 1) ? 0xffff0000u : 0u);
        *(u32x4*)(smem + LDS_BYTES + t * 16) = m;       // (256 threads)
        __syncthreads();
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int wc = wave / WP, wp = wave % WP;

    const int n_ctiles = a.Cout / BC;
    int ptile, ctile;
    const int HoWo = a.Ho * a.Wo;
    if (XMASK && a.lazy_order) {                      // (launcher: Ho Wo % BP == 0, N % in_mod == 0) the samples of one activation tile
        if (!lazy_tile_map(blockIdx.x, a.in_mod, a.N / a.in_mod, HoWo / BP, n_ctiles, ptile, ctile)) return;   // back to back on one XCD
    } else {
        xcd_tile_map(blockIdx.x, (a.M + BP - 1) / BP, n_ctiles, ptile, ctile);
    }
    const int ch0 = ctile * BC;
    const int pix0 = ptile * BP;

    const int Ktot = a.ksize * a.ksize * a.Cin;

    // ---- staging geometry (per thread: one 16-byte chunk of WROWS + XROWS rows) -------------
    const int chunk = tid & 7;
    const int r0 = tid >> 3;
    const int st_off = r0 * 128 + ((chunk ^ ((r0 >> 1) & 7)) << 4);
    const _Float16* wptr = a.wgt + (size_t)(ch0 + r0) * Ktot + chunk * 8;

    // Per staged pixel row: the tap-independent part of its input address — image base + the element offset of input pixel
    // (iy0, ix0) (negative inside the padding ring) — so that a K-step adds ONE launch-uniform offset ((ky W + kx) Cin + c0) per row
    // instead of recomputing (iy W + ix) Cin in 64 bits, twice with keep bits.  (rocprofv3 on ResNet-50's 256 -> 128 stride-2 reader:
    // waves issuing 41 % of their cycles at two per SIMD, MFMA-busy 0.29: the loop was bound by its address and mask arithmetic.)
    int iy0[XROWS], ix0[XROWS];
    const uint8_t* kbase[XMASK ? XROWS : 1];
    const _Float16* xbase[XROWS];
    bool vm[XROWS];
#pragma unroll
    for (int j = 0; j < XROWS; ++j) {
        const int m = pix0 + r0 + 32 * j;
        vm[j] = m < a.M;
        const int mm = vm[j] ? m : 0;
        const int nc = mm / HoWo;
        const int rem = mm - nc * HoWo;
        const int n = map_image<IMAP>(a, nc);
        const int oy = rem / a.Wo;
        const int ox = rem - oy * a.Wo;
        iy0[j] = oy * a.stride - a.pad;
        ix0[j] = ox * a.stride - a.pad;
        const int e0 = (iy0[j] * a.W + ix0[j]) * a.Cin;               // |e0| < H W Cin < 2^31 (launch_conv_igemm)
        xbase[j] = a.in + (size_t)(n % a.in_mod) * a.H * a.W * a.Cin + chunk * 8 + (ptrdiff_t)e0;
        // keep bits are per sample: the FOLDED image's rows; 8 elements per byte and e0 % 8 == 0
        if constexpr (XMASK) kbase[j] = a.in_bits + (size_t)n * a.H * a.W * (a.Cin >> 3) + chunk + (ptrdiff_t)(e0 >> 3);
    }

    // NOTE: staging registers are filled/drained by macros, not lambdas: with by-reference lambda
    // captures hipcc fails to scalarise the arrays and "promotes" them to LDS (an extra 16 KB and a
    // round trip through LDS per K-step).
    u32x4 wreg[WROWS], xreg[XROWS];
    uint32_t kreg[XMASK ? XROWS : 1];
#define GLOAD(KY, KX, C0)                                                                                  \
    {                                                                                                      \
        const int koff_ = ((KY) * a.ksize + (KX)) * a.Cin + (C0);                                          \
        const int toff_ = ((KY) * a.W + (KX)) * a.Cin + (C0);       /* launch-uniform */                   \
        _Pragma("unroll") for (int j = 0; j < WROWS; ++j)                                                  \
            wreg[j] = *(const u32x4*)(wptr + (size_t)(32 * j) * Ktot + koff_);                             \
        _Pragma("unroll") for (int j = 0; j < XROWS; ++j) {                                                \
            const int iy_ = iy0[j] + (KY), ix_ = ix0[j] + (KX);                                            \
            const bool ok_ = vm[j] && (unsigned)iy_ < (unsigned)a.H && (unsigned)ix_ < (unsigned)a.W;      \
            u32x4 v_ = {0u, 0u, 0u, 0u};                                                                   \
            if (ok_) v_ = *(const u32x4*)(xbase[j] + toff_);                                               \
            if constexpr (XMASK) {                                                                         \
                /* input-side MC-dropout: fetch the keep byte now, apply it at LSTORE time so the */       \
                /* loads stay in flight under the MFMAs of the current step */                             \
                uint32_t kb_ = 0;                                                                          \
                if (ok_) kb_ = kbase[j][toff_ >> 3];                                                       \
                kreg[j] = kb_;                                                                             \
            }                                                                                              \
            xreg[j] = v_;                                                                                  \
        }                                                                                                  \
    }
#define LSTORE(BUF)                                                                                        \
    {                                                                                                      \
        char* base_ = smem + (BUF) * TILE;                                                                 \
        _Pragma("unroll") for (int j = 0; j < WROWS; ++j) *(u32x4*)(base_ + st_off + j * 32 * 128) = wreg[j];   \
        _Pragma("unroll") for (int j = 0; j < XROWS; ++j) {                                                \
            u32x4 xv_ = xreg[j];                                                                           \
            if constexpr (XMASK) {   /* zero the dropped halves (1/(1-p) is folded into out_mul) */        \
                xv_ &= *(const u32x4*)(smem + LDS_BYTES + ((kreg[j] & 255u) << 4));                        \
            }                                                                                              \
            *(u32x4*)(base_ + BC * 128 + st_off + j * 32 * 128) = xv_;                                     \
        }                                                                                                  \
    }

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nK_all = a.ksize * a.ksize * (a.Cin / BK);
    int ks_begin = 0, nK = nK_all;
    int ky = 0, kx = 0, c0 = 0;
    if constexpr (SPLITK) {   // blockIdx.y takes K-steps [ks_begin, ks_begin + nK)
        ks_begin = (int)((long)blockIdx.y * nK_all / a.nsplit);
        nK = (int)((long)(blockIdx.y + 1) * nK_all / a.nsplit) - ks_begin;
        const int cpt = a.Cin / BK, tap = ks_begin / cpt;
        c0 = (ks_begin - tap * cpt) * BK;
        ky = tap / a.ksize;
        kx = tap - ky * a.ksize;
    }
    GLOAD(ky, kx, c0);
    if constexpr (DBUF) {
        LSTORE(0);
        __syncthreads();
    }

    const int sw_r = (r >> 1) & 7;
    const int a_row0 = wc * (BC / WC) + r;
    const int b_row0 = wp * (BP / WP) + r;

    for (int ks = 0; ks < nK; ++ks) {
        const int buf = DBUF ? (ks & 1) : 0;
        const bool more = ks + 1 < nK;
        if constexpr (!DBUF) {
            LSTORE(0);          // single buffer: registers -> LDS, then prefetch the next step into registers
            __syncthreads();
        }
        if (more) {
            c0 += BK;
            if (c0 == a.Cin) {
                c0 = 0;
                if (++kx == a.ksize) { kx = 0; ++ky; }
            }
            GLOAD(ky, kx, c0);
        }
        const char* wt = smem + buf * TILE;
        const char* xt = wt + BC * 128;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            const int coff = (((2 * kk + hh) ^ sw_r) << 4);
            half8 af[TI], bf[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = *(const half8*)(wt + (a_row0 + 32 * i) * 128 + coff);
#pragma unroll
            for (int j = 0; j < TJ; ++j) bf[j] = *(const half8*)(xt + (b_row0 + 32 * j) * 128 + coff);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = mfma_32x32x16<BF>(af[i], bf[j], acc[i][j]);
        }
        if constexpr (DBUF) {
            if (more) LSTORE(buf ^ 1);
        }
        __syncthreads();
    }

#undef GLOAD
#undef LSTORE
    // ---- epilogue ---------------------------------------------------------------------------
    if constexpr (SPLITK) {
        // raw partial sums; BN / ReLU / the 16-bit rounding happen in splitk_finish_kernel once all splits are added
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int m = pix0 + wp * (BP / WP) + 32 * j + r;
            if (m >= a.M) continue;
            float* prow = a.partial + ((size_t)blockIdx.y * a.M + m) * a.Cout;
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int c4 = ch0 + wc * (BC / WC) + 32 * i + 8 * g4 + 4 * hh;
                    *(float4*)(prow + c4) = make_float4(acc[i][j][4 * g4], acc[i][j][4 * g4 + 1], acc[i][j][4 * g4 + 2], acc[i][j][4 * g4 + 3]);
                }
        }
    } else if constexpr (BC == 128 && WC == 2 && WP == 2) {
        // coalesced through LDS (conv_epilogue.h); the double buffer is exactly the 64 KB it needs
        auto pixmap = [&](int p, int& n, int& rem) -> bool {
            const int m = pix0 + p;
            n = m / HoWo;
            rem = m - n * HoWo;
            n = map_image<IMAP>(a, n);
            return m < a.M;
        };
        auto offmap = [&](int p, size_t& off) -> bool {
            if constexpr (IMAP) {
                const int m = pix0 + p, n = m / HoWo;
                off = ((size_t)map_image<IMAP>(a, n) * HoWo + (m - n * HoWo)) * a.Cout;
            } else {
                off = (size_t)(pix0 + p) * a.Cout;
            }
            return pix0 + p < a.M;
        };
        epilogue_coalesced<TJ, PLAIN, 32, BF>(a, acc, smem, tid, ch0, pixmap, offmap);
    } else {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int m = pix0 + wp * (BP / WP) + 32 * j + r;
            if (m >= a.M) continue;
            const int n = m / HoWo;
            const PixelCtx px = make_pixel_ctx(a, map_image<IMAP>(a, n), m - n * HoWo);
#pragma unroll
            for (int i = 0; i < TI; ++i) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int c4 = ch0 + wc * (BC / WC) + 32 * i + 8 * g4 + 4 * hh;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g4 + e];
                    epilogue_quad<BF>(a, px, v, c4);
                }
            }
        }
    }
}

template <int BC, int BP, int WC, int WP, bool DBUF, bool XMASK = false>
static int launch_cfg(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    const int n_ctiles = a.Cout / BC;
    const long n_ptiles = ((long)a.M + BP - 1) / BP;
    const long blocks = n_ptiles * n_ctiles;
    if (blocks <= 0 || blocks > 0x7ffffff0L) return BMI_ERR_INVALID;
    a.lazy_order = XMASK && a.in_bits && opt_lazy_order() && !a.imap && a.in_mod < a.N && a.N % a.in_mod == 0 && (a.Ho * a.Wo) % BP == 0;
    const dim3 grid((unsigned)(a.lazy_order ? lazy_tile_grid(blocks) : blocks)), block(256);
    const bool plain = BC == 128 && conv_epilogue_is_plain(a);   // the 64-channel tiles use the per-quad epilogue: one instantiation
    if (a.imap) {   // dynamic early exit: the double-buffered 128-pixel configurations only
        if constexpr (DBUF && !XMASK && BP == 128) {
            if (a.bf16) {
                if (plain) hipLaunchKernelGGL((conv_igemm_kernel<BC, BP, WC, WP, DBUF, XMASK, BC == 128, true, true>), grid, block, 0, s, a);
                else hipLaunchKernelGGL((conv_igemm_kernel<BC, BP, WC, WP, DBUF, XMASK, false, true, true>), grid, block, 0, s, a);
            } else {
                if (plain) hipLaunchKernelGGL((conv_igemm_kernel<BC, BP, WC, WP, DBUF, XMASK, BC == 128, false, true>), grid, block, 0, s, a);
                else hipLaunchKernelGGL((conv_igemm_kernel<BC, BP, WC, WP, DBUF, XMASK, false, false, true>), grid, block, 0, s, a);
            }
        } else {
            return BMI_ERR_UNSUPPORTED;
        }
    } else if (a.bf16) {
        if (plain) hipLaunchKernelGGL((conv_igemm_kernel<BC, BP, WC, WP, DBUF, XMASK, BC == 128, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((conv_igemm_kernel<BC, BP, WC, WP, DBUF, XMASK, false, true>), grid, block, 0, s, a);
    } else {
        if (plain) hipLaunchKernelGGL((conv_igemm_kernel<BC, BP, WC, WP, DBUF, XMASK, BC == 128, false>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((conv_igemm_kernel<BC, BP, WC, WP, DBUF, XMASK, false, false>), grid, block, 0, s, a);
    }
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}

// out = fp16(relu?(bn(sum over the splits, in split order))): the plain epilogue's arithmetic on the added partial sums.
template <bool BF>
__global__ __launch_bounds__(256) void splitk_finish_kernel(ConvArgs a) {
    const long total = (long)a.M * (a.Cout >> 2);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % (a.Cout >> 2)) * 4;
        const long m = i / (a.Cout >> 2);
        float4 sum = *(const float4*)(a.partial + (size_t)m * a.Cout + c4);
        for (int sp = 1; sp < a.nsplit; ++sp) {
            const float4 p = *(const float4*)(a.partial + ((size_t)sp * a.M + m) * a.Cout + c4);
            sum.x += p.x; sum.y += p.y; sum.z += p.z; sum.w += p.w;
        }
        float v[4] = {sum.x, sum.y, sum.z, sum.w};
        half4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sc = (a.scale ? a.scale[c4 + e] : 1.f) * a.out_mul, bi = a.bias ? a.bias[c4 + e] : 0.f;
            float x = v[e] * sc + bi;
            if (a.relu) x = fmaxf(x, 0.f);
            o[e] = a16_from_f32<BF>(x);
        }
        *(half4*)(a.out + (size_t)m * a.Cout + c4) = o;
    }
}

int launch_splitk_finish(const ConvArgs& a, hipStream_t s) {
    if (!a.partial || a.nsplit < 2 || a.Cout % 4 != 0) return BMI_ERR_INVALID;
    const long total = (long)a.M * (a.Cout >> 2);
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (a.bf16) hipLaunchKernelGGL(splitk_finish_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(splitk_finish_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}

int launch_conv_igemm(const ConvArgs& a, hipStream_t s) {
    if (a.Cin % BK != 0 || a.Cout % 64 != 0 || a.in2) return BMI_ERR_UNSUPPORTED;   // the fused shortcut is a patch-kernel feature
    if (a.N <= 0 || a.M <= 0 || a.in_mod <= 0 || a.B <= 0) return BMI_ERR_INVALID;
    if (a.res && a.res_mod <= 0) return BMI_ERR_INVALID;
    if ((size_t)a.H * a.W * a.Cin >= 0x7fffffffull) return BMI_ERR_UNSUPPORTED;   // 31-bit in-image element offsets
    static const int big = [] { const char* v = std::getenv("BMI_IGEMM_BP256"); return v ? std::atoi(v) : 1; }();
    if (a.partial) {   // split-K: 128 x 128 tiles, nsplit workgroups per tile, then the finishing pass
        if (a.nsplit < 2 || a.nsplit > a.ksize * a.ksize * (a.Cin / BK) || !conv_epilogue_is_plain(a) || a.Cout % 128 != 0 || a.in_bits || a.imap)
            return BMI_ERR_INVALID;
        const long tiles = (((long)a.M + 127) / 128) * (a.Cout / 128);
        if (tiles > 0x7fffffffL) return BMI_ERR_INVALID;
        const dim3 grid((unsigned)tiles, (unsigned)a.nsplit), block(256);
        if (a.bf16) hipLaunchKernelGGL((conv_igemm_kernel<128, 128, 2, 2, true, false, true, true, false, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((conv_igemm_kernel<128, 128, 2, 2, true, false, true, false, false, true>), grid, block, 0, s, a);
        BMI_CHECK_LAUNCH();
        return launch_splitk_finish(a, s);
    }
    if (a.in_bits && a.imap) return BMI_ERR_UNSUPPORTED;   // (no dynamic-exit instantiation of the masked-input variant: the engine writes the masked tensor then)
    if (a.in_bits && a.lazy_planar) return BMI_ERR_UNSUPPORTED;   // the planar layout is conv3x3_s2's / conv3x3_patch's: a launch they refuse is materialised
    if (a.in_bits) {   // masked-input variant (register budget: 128-pixel tiles only)
        if (a.Cout % 128 == 0) return launch_cfg<128, 128, 2, 2, true, true>(a, s);
        return launch_cfg<64, 128, 1, 4, true, true>(a, s);
    }
    if (a.Cout % 128 == 0) {
        // 256-pixel tiles (single LDS buffer, 2 barriers per K-step) halve the weight-tile traffic per FLOP
        const long m_sel = a.n_ref > 0 ? (long)a.n_ref * a.Ho * a.Wo : a.M;
        if (big && !a.imap && (m_sel / 256) * (a.Cout / 128) >= 400 && a.ksize == 3) return launch_cfg<128, 256, 2, 2, false>(a, s);
        return launch_cfg<128, 128, 2, 2, true>(a, s);
    }
    // 64-channel convs (layer1 of the ResNets: B images of the once-per-batch prefix, ~1000 small workgroups): single LDS buffer,
    // 24 KB per workgroup -> six workgroups per CU instead of three: 59 / 65 / 55 / 64 -> 52 / 60 / 49 / 59 us for the four
    // launches of the headline net.  (A 64-channel instantiation of conv3x3_patch measured the same 220 us: these launches sit
    // on a ~20 us ramp + latency floor, not on their main loop.)  The dynamic-exit variants exist for the double-buffered form.
    if (!a.imap) return launch_cfg<64, 128, 1, 4, false>(a, s);
    return launch_cfg<64, 128, 1, 4, true>(a, s);
}
