// Wide-tile implicit-GEMM convolution for gfx950: 256 output channels x 256 pixels x 64 deep per K-step,
// 512 threads (8 waves), both operand tiles filled by LDS-DMA into a double-buffered 2 x 64 KB LDS ring.
//
// Why a second per-tap kernel: the stride-2 3x3 convs (BasicBlock conv1 of layers 2-4 and every exit-head
// conv, SA/models/resnet18/resnet18.py:280-299, :306-329) and the 1x1 convs of Bottleneck nets cannot keep
// an input patch in LDS (a stride-2 patch is 4x the bytes per output pixel), so every K-step streams an
// activation tile AND a weight tile from L2.  With conv_igemm's 128 x 256 tile that is 48 KB per 4.2 MFLOP,
// register-staged and single-buffered: measured 540-740 TFLOP/s on these shapes against 930-1030 for the
// patch kernel.  Here the tile is 256 x 256 (64 KB per 8.4 MFLOP: 1.5x fewer L2->LDS bytes per FLOP, half
// the barriers per FLOP), nothing is staged through VGPRs, and the next K-step's tiles are in flight
// while the current one computes (one barrier per K-step).
//
//   waves      = 2 channel halves (g) x [2 (channels) x 2 (pixels)]: each wave owns 64 ch x 128 px
//                (2 x 4 tiles of v_mfma_f32_32x32x16_f16), exactly the wave tile of conv_igemm<128,256>
//   LDS rows   = 128 B (64 fp16 of one weight row / one pixel); 16-byte chunk c of row r is stored at
//                chunk c ^ ((r >> 1) & 7) (conflict-free ds_read_b128); the DMA writes lane-linearly, so
//                the permutation is applied to the per-lane SOURCE address
//   padding    = out-of-image taps and tile rows beyond M are DMA'd from a zero page
//   epilogue   = each channel half runs conv_epilogue.h's coalesced epilogue in its own 64 KB of the ring
//   pair mode  = two convs that read the SAME input with the same geometry (layerN.0.conv1 and the first
//                conv of the exit head before it) run as one launch: channel tiles below `split` use the
//                first conv's weights / BN / output tensor, the others the second's.  The input tile is
//                then fetched once for both, and a 128-channel conv still fills the 256-channel tile.
#include <cstdlib>

#include "conv_epilogue.h"
#include "kernels.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static __device__ unsigned int g_zero_page_w[64];   // 256 B of zeros (DMA source for padding / tail rows)

#define GLDS16(SRC, LDSPTR)                                                                     \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC),       \
                                     (__attribute__((address_space(3))) void*)(LDSPTR), 16, 0, 0)

#ifdef BMI_WIDE_STAMPS
// Diagnostic build only (tools/ab_build.py stamps:-DBMI_WIDE_STAMPS): phase timestamps of wave 0 of each workgroup.
__device__ unsigned long long g_wide_stamps[8192 * 8];
#define STAMP(SLOT) if (tid == 0 && blockIdx.x < 8192) g_wide_stamps[blockIdx.x * 8 + (SLOT)] = __builtin_readcyclecounter();
#if BMI_WIDE_STAMPS >= 2   // per-K-step accumulators (each costs wave 0 a global round trip: inflates the main loop)
#define STAMP_ADD(SLOT, T0) if (tid == 0 && blockIdx.x < 8192) g_wide_stamps[blockIdx.x * 8 + (SLOT)] += __builtin_readcyclecounter() - (T0);
#define STAMP_T0(V) const unsigned long long V = __builtin_readcyclecounter();
#else
#define STAMP_ADD(SLOT, T0)
#define STAMP_T0(V)
#endif
extern "C" int bmi_debug_wide_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wide_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -5;
}
extern "C" int bmi_debug_wide_stamps_clear() {
    static unsigned long long z[8192 * 8];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_wide_stamps), z, sizeof(z)) == hipSuccess ? 0 : -5;
}
#else
#define STAMP(SLOT)
#define STAMP_ADD(SLOT, T0)
#define STAMP_T0(V)
#endif

#define WBC 256
#define WBP 256
#define WSTAGE ((WBC + WBP) * 128)

// MS = MFMA shape (32: v_mfma_f32_32x32x16_f16, 16: v_mfma_f32_16x16x32_f16), same wave tile and LDS traffic either way;
// chosen per launch (see conv3x3_patch.hip).
#define WIDE_SHAPE_CONSTS                                                                              \
    constexpr int TJ = 4;                                                                              \
    constexpr int TI = MS == 32 ? 2 : 4, TP = MS == 32 ? 4 : 8, RW = MS;                               \
    typedef float accv __attribute__((ext_vector_type(MS == 32 ? 16 : 4)));
#define WIDE_MFMA(AFR, BFR, ACC)                                                                       \
    if constexpr (MS == 32) ACC = mfma_32x32x16<BF>(AFR, BFR, ACC);                                    \
    else ACC = mfma_16x16x32<BF>(AFR, BFR, ACC);
// One phase of the ping-pong loop: LOAD part (fragment reads of k-substep / pixel half KK of the K-step in stage ST, plus
// this wave's share of the next K-step's DMA), barrier, MFMA part, barrier.  MS = 32: phase = one 16-deep k-substep,
// 2 + 4 reads, 8 MFMAs of 32 cycles.  MS = 16: phase = one pixel half (4 of the 8 pixel tiles) of a 32-deep k-substep,
// 4 (first half only: the channel fragments are kept for the second) + 4 reads, 16 MFMAs of 16 cycles.
#define WIDE_PHASE_LOAD(KK, ST)                                                                        \
    if constexpr (MS == 32) {                                                                          \
        const int coff = ((2 * (KK) + kq) ^ sw_r) << 4;                                                \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) af[i] = *(const half8*)((ST) + a_off + i * 32 * 128 + coff);  \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) bf[j] = *(const half8*)((ST) + b_off + j * 32 * 128 + coff);  \
    } else {                                                                                           \
        const int coff = ((4 * ((KK) >> 1) + kq) ^ sw_r) << 4;                                         \
        if (((KK) & 1) == 0) {                                                                         \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) af[i] = *(const half8*)((ST) + a_off + i * 16 * 128 + coff); \
        }                                                                                              \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) bf[j] = *(const half8*)((ST) + b_off + (4 * ((KK) & 1) + j) * 16 * 128 + coff); \
    }
#define WIDE_PHASE_MFMA(KK)                                                                            \
    if constexpr (MS == 32) {                                                                          \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                  \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) { WIDE_MFMA(af[i], bf[j], acc[i][j]); }      \
    } else {                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                  \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) { WIDE_MFMA(af[i], bf[j], acc[i][4 * ((KK) & 1) + j]); } \
    }

template <int EPI, int MS, bool BF, bool IMAP = false>
__global__ __launch_bounds__(512, 1) void conv_igemm_wide_kernel(ConvArgs a) {
    WIDE_SHAPE_CONSTS
    __shared__ __attribute__((aligned(16))) char smem[2 * WSTAGE];
    static_assert(2 * WSTAGE == 2 * BMI_EPILOGUE_LDS_BYTES, "one epilogue staging area per channel half");

    const int tid = threadIdx.x;
    STAMP(0);
#ifdef BMI_WIDE_STAMPS
    // the cycle counter (slots 0-3) is not comparable between workgroups; the 100 MHz real-time counter (slots 5, 6) is
    if (tid == 0 && blockIdx.x < 8192) g_wide_stamps[blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memrealtime();
#endif
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & (RW - 1), kq = lane / RW;
    const int g = wave >> 2, wc = (wave >> 1) & 1, wp = wave & 1;

    const int n_ctiles = a.Cout / WBC;
    int ptile, ctile;
    xcd_tile_map(blockIdx.x, (a.M + WBP - 1) / WBP, n_ctiles, ptile, ctile, a.xcd_split);
    const int ch0 = ctile * WBC;
    const int pix0 = ptile * WBP;
    const int HoWo = a.Ho * a.Wo;
    const int Ktot = a.ksize * a.ksize * a.Cin;
    const int split = a.wgt_b ? a.split : a.Cout;

    // ---- per-thread DMA sources: piece q = tid + 512*i -> tile row (q >> 3) = 64*i + (tid >> 3), slot tid & 7 ----
    const int rowt = tid >> 3;
    const int srcchunk = ((tid & 7) ^ ((rowt >> 1) & 7)) * 8;   // (64*i >> 1) & 7 == 0: the same for all i
    const _Float16* wsrc[4];
    const _Float16* xsrc[4];
    int iy0[4], ix0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = ch0 + 64 * i + rowt;
        wsrc[i] = (ch < split ? a.wgt + (size_t)ch * Ktot : a.wgt_b + (size_t)(ch - split) * Ktot) + srcchunk;
    }
    // the first weight tile is on its way while the pixel rows are decoded (divisions) and the accumulators cleared
#pragma unroll
    for (int i = 0; i < 4; ++i) GLDS16(wsrc[i], smem + (i * 512 + wave * 64) * 16);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = pix0 + 64 * i + rowt;
        const bool vm = m < a.M;
        const int mm = vm ? m : 0;
        const int n = mm / HoWo;
        const int rem = mm - n * HoWo;
        const int oy = rem / a.Wo;
        const int ox = rem - oy * a.Wo;
        iy0[i] = vm ? oy * a.stride - a.pad : -0x10000;       // a row beyond M never passes the bounds test
        ix0[i] = ox * a.stride - a.pad;
        xsrc[i] = a.in + (size_t)(map_image<IMAP>(a, n) % a.in_mod) * a.H * a.W * a.Cin + srcchunk;
    }
    // One row block (64 rows = 8 KB) of the weight / activation tile per call: the fills of the NEXT K-step are
    // spread over the MFMAs of the current one (2 DMA instructions per 8 MFMAs per wave).  Issued as one burst
    // after the barrier they cost ~960 cycles per K-step during which no wave of the workgroup issues an MFMA
    // (phase stamps, tools/wide_stamps.py): 64 KB per K-step at the CU's 64 B/clk vector-memory path.
#define ISSUE_W(I, KOFF, ST)  GLDS16(wsrc[I] + (KOFF), (ST) + ((I) * 512 + wave * 64) * 16)
#define ISSUE_X(I, KY, KX, C0, ST)                                                                     \
    {                                                                                                  \
        const int iy_ = iy0[I] + (KY), ix_ = ix0[I] + (KX);                                            \
        const bool ok_ = (unsigned)iy_ < (unsigned)a.H && (unsigned)ix_ < (unsigned)a.W;               \
        const _Float16* s_ = ok_ ? xsrc[I] + (size_t)(iy_ * a.W + ix_) * a.Cin + (C0)                  \
                                 : (const _Float16*)g_zero_page_w;                                     \
        GLDS16(s_, (ST) + WBC * 128 + ((I) * 512 + wave * 64) * 16);                                   \
    }

    {   // first activation tile (the weight tile was issued above)
        char* st0 = smem;
#pragma unroll
        for (int i = 0; i < 4; ++i) ISSUE_X(i, 0, 0, 0, st0);
    }
    __builtin_amdgcn_sched_barrier(0);
    accv acc[TI][TP];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int e = 0; e < (MS == 32 ? 16 : 4); ++e) acc[i][j][e] = 0.f;

    const int a_off = (g * 128 + wc * 64 + r) * 128;
    const int b_off = WBC * 128 + (wp * 128 + r) * 128;
    const int sw_r = (r >> 1) & 7;

    const int nK = a.ksize * a.ksize * (a.Cin / 64);
    int ky = 0, kx = 0, c0 = 0;
    // ---- ping-pong main loop ------------------------------------------------------------------------------------
    // A K-step is four phases (one 16-deep k-substep each); a phase is a LOAD part (6 ds_read_b128 of the substep's
    // fragments + this wave's share of the next K-step's LDS-DMA), a barrier, an MFMA part (8 MFMAs at raised
    // priority), a barrier.  The two channel halves (waves 0-3 / 4-7: one wave per SIMD each) run ONE BARRIER APART, so
    // on every SIMD one wave is in its MFMA part while the other reads LDS and issues DMA: the matrix pipe never
    // waits for a fragment read, and the barrier is the hand-over between the two waves, not an idle point.
    // (With all eight waves in lockstep the same loop spent 570 cycles per K-step at the barrier, 370 waiting for
    // the DMA and 2390 instead of 2048 in the MFMA phase: tools/wide_stamps.py.)
    //
    // Intervals between consecutive barriers, K-step T, phase k:  group 0: LOAD 8T+2k, MFMA 8T+2k+1
    //                                                              group 1: LOAD 8T+2k+1, MFMA 8T+2k+2
    // Raw s_barrier throughout (a __syncthreads() would drain the LDS-DMA: vmcnt(0)).  Hazards, by construction:
    //   WAR  the other buffer was last read for K-step T-1 by group 1 in interval 8T-1 (retired by its lgkmcnt(0) at
    //        the start of 8T): group 1 may refill it from 8T+1 (its phases 0,1), group 0 from 8T+2 (its phases 1,2).
    //   RAW  every wave retires its own DMA (vmcnt(0)) in interval 8T+7, i.e. before the barrier that ends 8T+7;
    //        the first reads of K-step T+1 are in 8T+8 (group 0) and 8T+9 (group 1).
    // Both groups execute the same number of barriers: group 1 one extra at the start, group 0 one extra at the end.
#define RAW_BARRIER()                                  \
    {                                                  \
        __builtin_amdgcn_sched_barrier(0);             \
        asm volatile("" ::: "memory");                 \
        __builtin_amdgcn_s_barrier();                  \
        asm volatile("" ::: "memory");                 \
        __builtin_amdgcn_sched_barrier(0);             \
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RAW_BARRIER();                      // K-step 0 has landed (every wave waited for its own pieces)
    STAMP(1);
    if (g == 1) RAW_BARRIER();          // stagger
    const int dma_phase = g == 0 ? 1 : 0;
    for (int ks = 0; ks < nK; ++ks) {
        const int buf = ks & 1;
        const bool more = ks + 1 < nK;
        if (more) {
            c0 += 64;
            if (c0 == a.Cin) {
                c0 = 0;
                if (++kx == a.ksize) { kx = 0; ++ky; }
            }
        }
        const int koff = (ky * a.ksize + kx) * a.Cin + c0;
        char* nst = smem + (buf ^ 1) * WSTAGE;
        const char* st = smem + buf * WSTAGE;
        half8 af[TI], bf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            // LOAD part
            WIDE_PHASE_LOAD(kk, st);
            if (more && kk == dma_phase) {
#pragma unroll
                for (int i = 0; i < 4; ++i) ISSUE_W(i, koff, nst);
            }
            if (more && kk == dma_phase + 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) ISSUE_X(i, ky, kx, c0, nst);
            }
            if (kk == 3 && g == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // interval 8T+7 (group 1: LOAD part)
            RAW_BARRIER();
            // MFMA part
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_setprio(1);
            WIDE_PHASE_MFMA(kk);
            __builtin_amdgcn_s_setprio(0);
            if (kk == 3 && g == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // interval 8T+7 (group 0: MFMA part)
            RAW_BARRIER();
        }
    }
    if (g == 0) RAW_BARRIER();          // re-align the two groups before the epilogue reuses the LDS
#undef RAW_BARRIER
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef ISSUE_W
#undef ISSUE_X
    STAMP(2);

    // ---- epilogue: each channel half (4 waves) in its own 64 KB ------------------------------------
    ConvArgs b = a;
    int chg = ch0 + 128 * g;
    if (a.wgt_b) {
        if (chg >= split) {
            b.out = a.out_b; b.scale = a.scale_b; b.bias = a.bias_b;
            b.Cout = a.Cout - split;
            chg -= split;
        } else {
            b.Cout = split;
        }
    }
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
            off = ((size_t)map_image<IMAP>(a, n) * HoWo + (m - n * HoWo)) * b.Cout;
        } else {
            off = (size_t)(pix0 + p) * b.Cout;
        }
        return pix0 + p < a.M;
    };
    epilogue_coalesced<TJ, EPI, MS, BF>(b, acc, smem + g * BMI_EPILOGUE_LDS_BYTES, tid & 255, chg, pixmap, offmap);
    STAMP(3);
#ifdef BMI_WIDE_STAMPS
    if (tid == 0 && blockIdx.x < 8192) g_wide_stamps[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent form of the kernel above for plain launches (BN + ReLU epilogue: every stride-2 conv of the path): one
// workgroup per CU walks over tiles blockIdx, blockIdx + gridDim, ...  What it buys: the first K-step of a tile is an
// HBM round trip (5.6k of a workgroup's 45-130k cycles, stamps) and a CU sits idle between the end of one workgroup and
// the start of the next (6.5 % of a launch).  Here the next tile's first K-step is issued into stage buffer 0 as soon
// as the main loop has released the buffers, and lands while the current tile's epilogue runs in stage buffer 1
// (fp16, two rounds of 128 pixels per channel half).  BN scale / bias live in a small LDS table: an ordinary global
// load in the epilogue would make hipcc drain the LDS-DMA (vmcnt(0)) at its first use.
#define WBN_MAX 1024

template <int MS, bool BF>
__global__ __launch_bounds__(512, 1) void conv_igemm_wide_persist_kernel(ConvArgs a, int n_tiles) {
    WIDE_SHAPE_CONSTS
    __shared__ __attribute__((aligned(16))) char smem[2 * WSTAGE + 2 * WBN_MAX * 4];
    float* const bn_scale = (float*)(smem + 2 * WSTAGE);
    float* const bn_bias = bn_scale + WBN_MAX;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & (RW - 1), kq = lane / RW;
    const int g = wave >> 2, wc = (wave >> 1) & 1, wp = wave & 1;
    const int n_ctiles = a.Cout / WBC;
    const int n_ptiles = (a.M + WBP - 1) / WBP;
    const int HoWo = a.Ho * a.Wo;
    const int Ktot = a.ksize * a.ksize * a.Cin;
    const int split = a.wgt_b ? a.split : a.Cout;

    for (int c = tid; c < a.Cout; c += 512) {
        const bool second = c >= split;
        const float* sp = second ? a.scale_b : a.scale;
        const float* bp = second ? a.bias_b : a.bias;
        const int cc = second ? c - split : c;
        bn_scale[c] = (sp ? sp[cc] : 1.f) * a.out_mul;
        bn_bias[c] = bp ? bp[cc] : 0.f;
    }
    __syncthreads();   // (no LDS-DMA in flight yet: the plain barrier and its waits are fine here)

    const int rowt = tid >> 3;
    const int srcchunk = ((tid & 7) ^ ((rowt >> 1) & 7)) * 8;
    const int a_off = (g * 128 + wc * 64 + r) * 128;
    const int b_off = WBC * 128 + (wp * 128 + r) * 128;
    const int sw_r = (r >> 1) & 7;
    const int nK = a.ksize * a.ksize * (a.Cin / 64);
    const int dma_phase = g == 0 ? 1 : 0;
    const _Float16* wsrc[4];
    const _Float16* xsrc[4];
    int iy0[4], ix0[4];
    int ch0 = 0, pix0 = 0;

#define ISSUE_W(I, KOFF, ST)  GLDS16(wsrc[I] + (KOFF), (ST) + ((I) * 512 + wave * 64) * 16)
#define ISSUE_X(I, KY, KX, C0, ST)                                                                     \
    {                                                                                                  \
        const int iy_ = iy0[I] + (KY), ix_ = ix0[I] + (KX);                                            \
        const bool ok_ = (unsigned)iy_ < (unsigned)a.H && (unsigned)ix_ < (unsigned)a.W;               \
        const _Float16* s_ = ok_ ? xsrc[I] + (size_t)(iy_ * a.W + ix_) * a.Cin + (C0)                  \
                                 : (const _Float16*)g_zero_page_w;                                     \
        GLDS16(s_, (ST) + WBC * 128 + ((I) * 512 + wave * 64) * 16);                                   \
    }
    // tile VB: DMA sources of this thread's 4 weight rows and 4 pixel rows, and its first K-step -> stage buffer 0
#define SETUP_TILE(VB)                                                                                 \
    {                                                                                                  \
        int ptile_, ctile_;                                                                            \
        xcd_tile_map((VB), n_ptiles, n_ctiles, ptile_, ctile_, a.xcd_split);                           \
        ch0 = ctile_ * WBC;                                                                            \
        pix0 = ptile_ * WBP;                                                                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                \
            const int ch = ch0 + 64 * i + rowt;                                                        \
            wsrc[i] = (ch < split ? a.wgt + (size_t)ch * Ktot : a.wgt_b + (size_t)(ch - split) * Ktot) + srcchunk; \
        }                                                                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) GLDS16(wsrc[i], smem + (i * 512 + wave * 64) * 16);           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                \
            const int m = pix0 + 64 * i + rowt;                                                        \
            const bool vm = m < a.M;                                                                   \
            const int mm = vm ? m : 0;                                                                 \
            const int n = mm / HoWo;                                                                   \
            const int rem = mm - n * HoWo;                                                             \
            const int oy = rem / a.Wo;                                                                 \
            const int ox = rem - oy * a.Wo;                                                            \
            iy0[i] = vm ? oy * a.stride - a.pad : -0x10000;                                            \
            ix0[i] = ox * a.stride - a.pad;                                                            \
            xsrc[i] = a.in + (size_t)(n % a.in_mod) * a.H * a.W * a.Cin + srcchunk;                    \
        }                                                                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) ISSUE_X(i, 0, 0, 0, smem);                       \
    }
#define RAW_BARRIER()                                  \
    {                                                  \
        __builtin_amdgcn_sched_barrier(0);             \
        asm volatile("" ::: "memory");                 \
        __builtin_amdgcn_s_barrier();                  \
        asm volatile("" ::: "memory");                 \
        __builtin_amdgcn_sched_barrier(0);             \
    }

    int vb = blockIdx.x;
    SETUP_TILE(vb);
    while (vb < n_tiles) {
        const int cur_ch0 = ch0, cur_pix0 = pix0;
        accv acc[TI][TP];
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j)
#pragma unroll
                for (int e = 0; e < (MS == 32 ? 16 : 4); ++e) acc[i][j][e] = 0.f;

        // ---- ping-pong main loop (see conv_igemm_wide_kernel for the interval / hazard table) ----
        int ky = 0, kx = 0, c0 = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile's first K-step (and the previous tile's stores)
        RAW_BARRIER();
        if (g == 1) RAW_BARRIER();
        for (int ks = 0; ks < nK; ++ks) {
            const int buf = ks & 1;
            const bool more = ks + 1 < nK;
            if (more) {
                c0 += 64;
                if (c0 == a.Cin) {
                    c0 = 0;
                    if (++kx == a.ksize) { kx = 0; ++ky; }
                }
            }
            const int koff = (ky * a.ksize + kx) * a.Cin + c0;
            char* nst = smem + (buf ^ 1) * WSTAGE;
            const char* st = smem + buf * WSTAGE;
            half8 af[TI], bf[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                WIDE_PHASE_LOAD(kk, st);
                if (more && kk == dma_phase) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) ISSUE_W(i, koff, nst);
                }
                if (more && kk == dma_phase + 1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) ISSUE_X(i, ky, kx, c0, nst);
                }
                if (kk == 3 && g == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                RAW_BARRIER();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_setprio(1);
                WIDE_PHASE_MFMA(kk);
                __builtin_amdgcn_s_setprio(0);
                if (kk == 3 && g == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                RAW_BARRIER();
            }
        }
        if (g == 0) RAW_BARRIER();      // both groups aligned: every wave is done with both stage buffers

        // ---- next tile's first K-step -> stage buffer 0; it lands during the epilogue below ----
        const int nvb = vb + (int)gridDim.x;
        if (nvb < n_tiles) SETUP_TILE(nvb);

        // ---- epilogue of the current tile in stage buffer 1: 32 KB per channel half, two rounds of 128 pixels ----
        {
            char* const E = smem + WSTAGE + g * 32768;
            const int tl = tid & 255;
            const int chl = cur_ch0 + 128 * g;              // launch-wide channel of this half's channel 0 (BN table index)
            _Float16* outp = a.out;
            int oc = a.Cout, chg = chl;
            if (a.wgt_b) {
                if (chl >= split) { outp = a.out_b; oc = a.Cout - split; chg = chl - split; }
                else oc = split;
            }
            const int k = tl & 15;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                if (rr) lds_barrier();                      // round 0's reads are done
                if constexpr (MS == 16) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int c4 = chl + wc * 64 + 16 * i + 4 * kq;
                        const f32x4_e sc = *(const f32x4_e*)(bn_scale + c4), bi = *(const f32x4_e*)(bn_bias + c4);
                        const int cq = wc * 8 + 2 * i + (kq >> 1);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const int p = wp * 64 + jj * 16 + r;                     // pixel inside the round
                            half4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float v = acc[i][4 * rr + jj][e] * sc[e] + bi[e];
                                if (a.relu) v = fmaxf(v, 0.f);
                                o[e] = a16_from_f32<BF>(v);
                            }
                            *(half4*)(E + p * 256 + ((cq ^ r) << 4) + (((kq ^ jj) & 1) << 3)) = o;
                        }
                    }
                } else {
                const int hh = kq, hsw = hh ^ ((r >> 4) & 1);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int i = q >> 2, g4 = q & 3;
                    const int c4 = chl + wc * 64 + 32 * i + 8 * g4 + 4 * hh;
                    const f32x4_e sc = *(const f32x4_e*)(bn_scale + c4), bi = *(const f32x4_e*)(bn_bias + c4);
                    const int cq = wc * 8 + 4 * i + g4;
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int p = wp * 64 + jj * 32 + r;                     // pixel inside the round
                        half4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[i][2 * rr + jj][4 * g4 + e] * sc[e] + bi[e];
                            if (a.relu) v = fmaxf(v, 0.f);
                            o[e] = a16_from_f32<BF>(v);
                        }
                        *(half4*)(E + p * 256 + ((cq ^ (r & 15)) << 4) + hsw * 8) = o;
                    }
                }
                }
                lds_barrier();
                half8_e o8[8];
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int pl = (tl >> 4) + 16 * it;
                    o8[it] = *(const half8_e*)(E + pl * 256 + ((k ^ (pl & 15)) << 4));
                }
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int pl = (tl >> 4) + 16 * it;
                    const int m = cur_pix0 + (pl >> 6) * (32 * TJ) + rr * 64 + (pl & 63);
                    if (m >= a.M) continue;
                    half8_e v = o8[it];
                    if (it & 1) v = __builtin_shufflevector(v, v, 4, 5, 6, 7, 0, 1, 2, 3);
                    *(half8_e*)(outp + (size_t)m * oc + chg + 8 * k) = v;
                }
            }
        }
        vb = nvb;
    }
#undef ISSUE_W
#undef ISSUE_X
#undef SETUP_TILE
#undef RAW_BARRIER
}

// Shapes the wide kernel takes.  `cout` is the launch's total channel count (both convs of a pair).
bool conv_takes_wide_kernel(int cin, int cout) {
    static const int on = [] { const char* v = std::getenv("BMI_IGEMM_WIDE"); return v ? std::atoi(v) : 1; }();
    return on && cin % 64 == 0 && cout % WBC == 0;
}

int launch_conv_igemm_wide(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    a.xcd_split = xcd_split_for(a.Cout / WBC, (size_t)a.Cout * a.ksize * a.ksize * a.Cin * 2);
    if (!conv_takes_wide_kernel(a.Cin, a.Cout) || a.in2 || a.in_bits) return BMI_ERR_UNSUPPORTED;
    if (a.N <= 0 || a.M <= 0 || a.in_mod <= 0 || a.B <= 0 || (a.res && a.res_mod <= 0)) return BMI_ERR_INVALID;
    if (a.wgt_b) {
        if (!a.out_b || a.split <= 0 || a.split >= a.Cout || a.split % 128 != 0 || a.res || a.site.kind != BMI_SITE_NONE)
            return BMI_ERR_INVALID;
    }
    const long blocks = (((long)a.M + WBP - 1) / WBP) * (a.Cout / WBC);
    if (blocks > 0x7fffffffL) return BMI_ERR_INVALID;
    static const int n_cu = [] {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cu = 0;
        return cu;
    }();
    // a grid that cannot fill the chip (small deterministic-prefix launches: B images, not samples x B) is better served by
    // conv_igemm's 128 x 128 tiles (4x as many workgroups): below 3/4 of a workgroup per CU (BMI_WIDE_MIN_BLOCKS overrides)
    static const int min_blocks = [] { const char* v = std::getenv("BMI_WIDE_MIN_BLOCKS"); return v ? std::atoi(v) : (n_cu > 0 ? 3 * n_cu / 4 : 192); }();
    const long blocks_sel = a.n_ref > 0 ? (((long)a.n_ref * a.Ho * a.Wo + WBP - 1) / WBP) * (a.Cout / WBC) : blocks;
    if (!a.wgt_b && blocks_sel < min_blocks) return BMI_ERR_UNSUPPORTED;
    static const int persist = [] { const char* v = std::getenv("BMI_WIDE_PERSIST"); return v ? std::atoi(v) : 1; }();
    const int shape = opt_mfma_shape_wide();
    if (persist && !a.imap && n_cu > 0 && conv_epilogue_is_plain(a) && a.Cout <= WBN_MAX && blocks * 10 > (long)opt_wide_persist_min() * n_cu) {
        if (a.bf16) hipLaunchKernelGGL((conv_igemm_wide_persist_kernel<16, true>), dim3((unsigned)n_cu), dim3(512), 0, s, a, (int)blocks);
        else if (shape == 16) hipLaunchKernelGGL((conv_igemm_wide_persist_kernel<16, false>), dim3((unsigned)n_cu), dim3(512), 0, s, a, (int)blocks);
        else hipLaunchKernelGGL((conv_igemm_wide_persist_kernel<32, false>), dim3((unsigned)n_cu), dim3(512), 0, s, a, (int)blocks);
        BMI_CHECK_LAUNCH();
        return BMI_OK;
    }
    const dim3 grid((unsigned)blocks), block(512);
    const int ms = (a.imap || a.bf16) ? 16 : shape;   // bf16 / dynamic early exit: the 16x16x32 instantiations only
    const int epi = opt_epilogue_lite() ? conv_epilogue_kind(a, ms) : (conv_epilogue_is_plain(a) ? BMI_EPI_PLAIN : BMI_EPI_GENERAL);
#define WIDE_LAUNCH(EPI_, MS_, BF_, IMAP_) hipLaunchKernelGGL((conv_igemm_wide_kernel<EPI_, MS_, BF_, IMAP_>), grid, block, 0, s, a)
#define WIDE_LAUNCH_EPI16(BF_, IMAP_)                                            \
    {                                                                            \
        if (epi == BMI_EPI_PLAIN) WIDE_LAUNCH(BMI_EPI_PLAIN, 16, BF_, IMAP_);    \
        else if (epi == BMI_EPI_LITE) WIDE_LAUNCH(BMI_EPI_LITE, 16, BF_, IMAP_); \
        else WIDE_LAUNCH(BMI_EPI_GENERAL, 16, BF_, IMAP_);                       \
    }
    if (a.imap) {
        if (a.bf16) WIDE_LAUNCH_EPI16(true, true)
        else WIDE_LAUNCH_EPI16(false, true)
    } else if (a.bf16) {
        WIDE_LAUNCH_EPI16(true, false)
    } else if (ms == 16) {
        WIDE_LAUNCH_EPI16(false, false)
    } else {
        if (epi == BMI_EPI_PLAIN) WIDE_LAUNCH(BMI_EPI_PLAIN, 32, false, false);
        else WIDE_LAUNCH(BMI_EPI_GENERAL, 32, false, false);
    }
#undef WIDE_LAUNCH_EPI16
#undef WIDE_LAUNCH
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}
