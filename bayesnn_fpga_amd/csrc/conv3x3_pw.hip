// 3x3 / stride-1 / pad-1 convolution on small maps (8x8, 4x4) with Cout % 256 == 0: the "patch-wide" kernel — the input
// patch resident in LDS like conv3x3_patch, the 256 x 256 tile, 8 waves and ping-pong main loop of conv_igemm_wide.
//
// Why a third 3x3 kernel.  On the 8x8 / 4x4 maps conv3x3_patch's workgroup is 128 channels x 128 pixels, wave tile
// 64 x 64: 8 fragment reads (ds_read_b128) per 16 MFMAs.  Two such workgroups per CU read 128 KB of LDS per 64-deep K-step
// against 1024 cycles of MFMA issue per SIMD, i.e. the LDS pipe (128 B/clk) is as busy as the matrix pipe.  A build that
// skipped half of the pixel-fragment reads (wrong results, timing probe) ran the 8x8 and 4x4 classes 5-6 % faster, the
// 16x16 class (wave tile 64 x 128 already) not at all.  Here every wave owns 64 channels x 128 pixels (12 reads per 32 MFMAs),
// the weight tile is shared by 256 pixels instead of 128 and the patch by 256 channels instead of 128: half the L2 -> LDS
// bytes per FLOP.  That tile needs one 512-thread workgroup per CU, so nothing else covers a refill of the patch: it is
// double buffered in 32-channel sub-patches (K-step = one tap x 32 channels = one 16x16x32 MFMA per tile), and the next
// sub-patch streams in piece by piece behind the MFMAs of the current one.
//
//   tile       = IMGS whole images (256 pixels: 4 maps of 8x8 or 16 maps of 4x4) x 256 output channels
//   waves      = 2 channel halves (g) x [2 (channels) x 2 (pixels)], wave tile 64 ch x 128 px = 4 x 8 tiles of
//                v_mfma_f32_16x16x32 — accumulator layout and epilogue identical to conv_igemm_wide
//   LDS        = 3 weight stages [256 ch][32 k] (64-byte rows) + 2 sub-patches [cells][32 ch] (64-byte cells)
//   64-B rows  = four rows share a 256-byte bank window.  ds_read_b128 is served in the four lane groups {0-3, 12-15, 20-27},
//                {4-11, 16-19, 28-31}, {32-35, 44-47, 52-59}, {36-43, 48-51, 60-63} (MI355X_MICROARCH.md, LDS): with lane =
//                row + 16 kq a group reads chunk kq of rows 0-3 and 12-15 of a 16-row fragment and chunk kq ^ 1 of rows 4-11.
//                The 16-byte chunk c of weight row r is stored at position c ^ 2 ((r >> 2) & 1), chunk c of a patch cell in
//                patch row y at c ^ 2 (y & 1), the 16 pixels of an MFMA tile being a 4 x 4 block: every group then hits 16
//                distinct 16-byte slots, for every tap shift.  (Round 2 rotated the chunks by r >> 2 / y, which is
//                conflict-free for groups of 16 CONSECUTIVE lanes, not for the real ones: rocprofv3 SQ_LDS_BANK_CONFLICT =
//                49.5 % of SQ_LDS_IDX_ACTIVE on this kernel, every fragment read took 8 LDS cycles instead of 4 —
//                profiles/experiments/r3_lds_bank_conflicts_before.txt, tools/experiments/lds_bank.hip variants 20-24.)
//                LDS-DMA writes lane-linearly, so the permutation is applied to the per-lane SOURCE address.
//   main loop  = conv_igemm_wide's ping-pong (two wave groups one barrier apart, raw s_barrier, LOAD part / MFMA part),
//                two phases per K-step: phase 0 reads the 4 channel fragments + pixel tiles 0-3 and issues the weight DMA
//                of the step AFTER NEXT (a 32-deep K-step is too short to hide an L2 round trip: with 2 stages the
//                kernel ran 845-1030 TF/s) + one piece of the next sub-patch, phase 1 reads pixel tiles 4-7.  The 9 taps of a
//                chunk are unrolled: stage index, tap shift and vmcnt immediates are compile-time.
//                Intervals between barriers, K-step T: group 0 LOAD 4T+2k, MFMA 4T+2k+1; group 1 one later.
//                WAR  weight stage (T+2)%3 was last read in 4T-3 (group 1, phase 0 of T-1; retired by its lgkmcnt(0) in 4T-2):
//                     refilled from 4T on.  Sub-patch (c+1)&1 was last read in step 9c-1; its refill starts in step 9c+1.
//                RAW  in interval 4T+3 every wave waits (counted vmcnt) until only the DMA issued in step T and the patch
//                     piece of step T-1 are in flight, i.e. until the weights of step T+1 have landed, before the barrier that
//                     ends 4T+3; first reads of step T+1 in 4T+4.  At tap 8 the count (2) also retires the whole next sub-patch.
// Reference semantics: BasicBlock.forward SA/models/resnet18/resnet18.py:32-48 (conv2 of every block, conv1 of the
// stride-1 blocks).
#include <cstdlib>

#include "conv_epilogue.h"
#include "kernels.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

static __device__ unsigned int g_zero_page_pw[64];

#define GLDS16(SRC, LDSPTR)                                                                     \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC),       \
                                     (__attribute__((address_space(3))) void*)(LDSPTR), 16, 0, 0)

// TW = map size (TW x TW): 8 or 4.  (A 128-channel x 512-pixel variant of this kernel for the 16x16 / Cout = 128 class was built
// and measured: 1061 vs 1094 TFLOP/s for conv3x3_patch, whose wave tile is 64 x 128 there already — not kept.)
template <int TW>
struct PwGeom {
    static constexpr int CT = 256, TH = TW, PX = 65536 / CT, IMGS = PX / (TH * TW);
    static_assert(IMGS >= 1 && IMGS * TH * TW == PX, "whole images per tile");
    static constexpr int PH = TH + 2, PW = TW + 2;
    static constexpr int PWP = (PW + 3) & ~3;                     // cell pitch, % 4 == 0: bank window of a cell = x & 3
    static constexpr int CELLS = IMGS * PH * PWP;                 // 8x8: 480, 4x4: 768, 16x16 (2 images): 720
    static constexpr int ITER_P = (CELLS * 4 + 511) / 512;        // 16-byte pieces per thread and sub-patch: 4 | 6 | 6
    static constexpr int PBUF = ITER_P * 512 * 16;                // 32 KB | 48 KB | 48 KB
    static constexpr int WROWS = CT / 128;                        // weight pieces per thread and stage
    static constexpr int WST = CT * 64;                           // one weight stage
    static constexpr int NST = 3;                                 // weight stages: the tile of K-step T+2 is in flight during T
    static constexpr int MAIN = NST * WST + 2 * PBUF;             // 112 KB | 144 KB | 120 KB
    static constexpr int SHORT = 2 * (WST + PX * 64);             // shortcut steps: two [weights | pixels] stages
    static constexpr int NEED = MAIN > SHORT ? MAIN : SHORT;
    static constexpr int LDS_BYTES = NEED > 2 * BMI_EPILOGUE_LDS_BYTES ? NEED : 2 * BMI_EPILOGUE_LDS_BYTES;
    // tile pixel p -> image of the tile and output coordinates: the 16 pixels of one MFMA tile are a 4 x 4 block
    __host__ __device__ static constexpr int p_img(int p) { return p / (TH * TW); }
    __host__ __device__ static constexpr int p_ox(int p) { return 4 * ((p >> 4) % (TW / 4)) + (p & 3); }
    __host__ __device__ static constexpr int p_oy(int p) { return 4 * (((p >> 4) / (TW / 4)) % (TH / 4)) + ((p >> 2) & 3); }
    // patch cell of the j-th 4 x 4 block of a wave (its 8 blocks start at a multiple of 8) relative to the wave's first one
    static constexpr int BR = TW / 4, BI = BR * BR;
    __host__ __device__ static constexpr int cell_delta(int j) { return (j / BI) * PH * PWP + 4 * ((j % BI) / BR) * PWP + 4 * ((j % BI) % BR); }
};

template <int TW, int EPI, bool BF, bool IMAP, bool POOL = false>
__global__ __launch_bounds__(512, 1) void conv3x3_pw_kernel(ConvArgs a) {
    static_assert(!POOL || (TW == 4 && EPI == BMI_EPI_LITE), "pooled output: 4x4 maps, the lite epilogue");
    using G = PwGeom<TW>;
    constexpr int CT = G::CT, TH = G::TH, IMGS = G::IMGS, PH = G::PH, PW = G::PW, PWP = G::PWP;
    constexpr int TJ = 4, TI = 4, TP = 8;
    typedef float accv __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) char smem[G::LDS_BYTES];
    char* const wst = smem;
    char* const pbuf = smem + G::NST * G::WST;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, kq = lane >> 4;
    const int g = wave >> 2, wc = (wave >> 1) & 1, wp = wave & 1;

    const int n_ctiles = a.Cout / CT;
    const int n_ptiles = (a.N + IMGS - 1) / IMGS;
    int ptile, ctile;
    xcd_tile_map(blockIdx.x, n_ptiles, n_ctiles, ptile, ctile, a.xcd_split);
    const int ch0 = ctile * CT;
    const int n0 = ptile * IMGS;
    const int Ktot = 9 * a.Cin;
    const int nC = a.Cin / 32;

    // ---- per-thread DMA sources ----
    // weights: piece q = tid + 512 i -> row (tid >> 2) + 128 i, position tid & 3 holds chunk pos ^ 2 ((row >> 2) & 1)
    const _Float16* wsrc[G::WROWS];
#pragma unroll
    for (int i = 0; i < G::WROWS; ++i) {
        const int row = (tid >> 2) + 128 * i;
        wsrc[i] = a.wgt + (size_t)(ch0 + row) * Ktot + ((tid & 3) ^ (((row >> 2) & 1) << 1)) * 8;
    }
#define ISSUE_W(KOFF, ST)                                                                        \
    {                                                                                            \
        _Pragma("unroll") for (int i = 0; i < G::WROWS; ++i)                                     \
            GLDS16(wsrc[i] + (KOFF), wst + (ST) * G::WST + (i * 512 + wave * 64) * 16);          \
    }
#define ISSUE_W_HALF(KOFF, ST, I) GLDS16(wsrc[I] + (KOFF), wst + (ST) * G::WST + ((I) * 512 + wave * 64) * 16)
#ifndef PW_ABL_HALFREADS
#define PW_ABL_HALFREADS 0   // timing probe (wrong results): every second pixel-fragment read is skipped (8 instead of 12 reads per 32 MFMAs)
#endif
#ifndef PW_WSPLIT
#define PW_WSPLIT 0          // 1: a wave's two weight DMA instructions of a K-step go out in the two phases (one each)
#endif
    ISSUE_W(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // sub-patch: piece q = tid + 512 i -> cell q >> 2, position q & 3 holds chunk pos ^ 2 (y & 1); element offset of its
    // source at channel chunk 0, or -1 (padding / halo beyond the map / cells of the pitch / images beyond N -> zero page)
    int psrc[G::ITER_P];
#pragma unroll
    for (int i = 0; i < G::ITER_P; ++i) {
        const int q = tid + 512 * i;
        const int cell = q >> 2, pos = q & 3;
        const int rowc = cell / PWP, x = cell - rowc * PWP;
        const int img = rowc / PH, y = rowc - img * PH;
        const int n = n0 + img;
        const int iy = y - 1, ix = x - 1;
        const bool ok = cell < G::CELLS && x < PW && n < a.N && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        psrc[i] = ok ? (int)((((size_t)(map_image<IMAP>(a, n) % a.in_mod) * a.H + iy) * a.W + ix) * a.Cin + (pos ^ ((y & 1) << 1)) * 8) : -1;
    }
#define ISSUE_P(I, C0, PB)                                                                                   \
    GLDS16(psrc[I] >= 0 ? a.in + (size_t)(unsigned)psrc[I] + (C0) : (const _Float16*)g_zero_page_pw,       \
           pbuf + (PB) * G::PBUF + ((I) * 512 + wave * 64) * 16)
#pragma unroll
    for (int i = 0; i < G::ITER_P; ++i) ISSUE_P(i, 0, 0);
    if (9 * nC > 1) ISSUE_W(a.Cin, 1);                         // K-step 1 = tap 1 of chunk 0
    __builtin_amdgcn_sched_barrier(0);

    // ---- per-lane fragment geometry ----
    const int a_off = (g * 128 + wc * 64 + l16) * 64;
    const int pbase = wp * 128;                                   // first tile pixel of this wave
    const int a_byte = (kq ^ (((l16 >> 2) & 1) << 1)) << 4;       // position of chunk kq in this lane's rows ((row >> 2) & 1 == (l16 >> 2) & 1)
    // Pixel fragments: cell of (wave's first block) + compile-time cell_delta(j) + this lane's cell inside the 4 x 4 block; the
    // chunk position kq ^ 2 (y & 1) depends on the lane's row in the block and on the tap's ky only (block origins are multiples
    // of 4) — so a fragment read is ONE ds_read_b128 with an immediate offset from one of three per-lane addresses.
    const int wave_cell = (G::p_img(pbase) * PH + G::p_oy(pbase)) * PWP + G::p_ox(pbase);
    int boff[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
        boff[ky] = (wave_cell + (l16 >> 2) * PWP + (l16 & 3)) * 64 + ((kq ^ ((((l16 >> 2) + ky) & 1) << 1)) << 4);

    accv acc[TI][TP];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

#define RAW_BARRIER()                                  \
    {                                                  \
        __builtin_amdgcn_sched_barrier(0);             \
        asm volatile("" ::: "memory");                 \
        __builtin_amdgcn_s_barrier();                  \
        asm volatile("" ::: "memory");                 \
        __builtin_amdgcn_sched_barrier(0);             \
    }
#define WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
    // What may still be in flight when K-step `TAP` of a chunk ends: the weight tile of the step after next (WROWS DMA
    // instructions, issued this step) and the sub-patch pieces issued this step and the step before (one each at taps
    // 1..ITER_P when another chunk follows); everything older — in particular the next step's weights — has landed.
#define END_OF_STEP_WAIT(TAP)                                                                                  \
    {                                                                                                          \
        constexpr int pieces_ = ((TAP) >= 1 && (TAP) <= G::ITER_P ? 1 : 0) + ((TAP) >= 2 && (TAP) <= G::ITER_P + 1 ? 1 : 0); \
        if (!last) { WAIT_VM(G::WROWS + pieces_); }                                                            \
        else if ((TAP) >= 7) { WAIT_VM(0); }                                                                   \
        else { WAIT_VM(G::WROWS); }                                                                            \
    }
    // One K-step = tap TAP x this chunk's 32 channels.  Two phases (LOAD part, barrier, MFMA part, barrier); the two wave
    // groups run one barrier apart.  Weight stage = TAP % 3 (9 taps per chunk: the stage index repeats per chunk).
#define PW_STEP(TAP)                                                                                           \
    {                                                                                                          \
        constexpr int ky_ = (TAP) / 3, kx_ = (TAP) - 3 * ky_;                                                  \
        const char* ws_ = wst + ((TAP) % 3) * G::WST + a_off + a_byte;                                         \
        const char* pb_ = pb + (ky_ * PWP + kx_) * 64;                                                         \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                     \
            if (kk == 0) {                                                                                     \
                _Pragma("unroll") for (int i = 0; i < TI; ++i) af[i] = *(const half8*)(ws_ + i * 16 * 64);     \
            }                                                                                                  \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                      \
                if (!PW_ABL_HALFREADS || !(j & 1)) bf[j] = *(const half8*)(pb_ + boff[ky_] + G::cell_delta(4 * kk + j) * 64); \
                else bf[j] = bf[j - 1];                                                                        \
            if (kk == 0 && !PW_WSPLIT) {                                                                       \
                /* weights of the step after next: (tap + 2) of this chunk, or taps 0 / 1 of the next one */   \
                if ((TAP) < 7) { ISSUE_W(((TAP) + 2) * a.Cin + c32, ((TAP) + 2) % 3); }                        \
                else if (!last) { ISSUE_W(((TAP) - 7) * a.Cin + c32 + 32, ((TAP) + 2) % 3); }                  \
                if ((TAP) >= 1 && (TAP) <= G::ITER_P && !last) { ISSUE_P((TAP) - 1, c32 + 32, nb); }           \
            }                                                                                                  \
            if (PW_WSPLIT) {                                                                                   \
                if ((TAP) < 7) { ISSUE_W_HALF(((TAP) + 2) * a.Cin + c32, ((TAP) + 2) % 3, kk); }               \
                else if (!last) { ISSUE_W_HALF(((TAP) - 7) * a.Cin + c32 + 32, ((TAP) + 2) % 3, kk); }         \
                if (kk == 1 && (TAP) >= 1 && (TAP) <= G::ITER_P && !last) { ISSUE_P((TAP) - 1, c32 + 32, nb); } \
            }                                                                                                  \
            if (kk == 1 && g == 1) END_OF_STEP_WAIT(TAP);      /* interval 4T+3, group 1: LOAD part */         \
            RAW_BARRIER();                                                                                     \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
            __builtin_amdgcn_s_setprio(1);                                                                     \
            _Pragma("unroll") for (int i = 0; i < TI; ++i)                                                     \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                  \
                    acc[i][4 * kk + j] = mfma_16x16x32<BF>(af[i], bf[j], acc[i][4 * kk + j]);                  \
            __builtin_amdgcn_s_setprio(0);                                                                     \
            if (kk == 1 && g == 0) END_OF_STEP_WAIT(TAP);      /* interval 4T+3, group 0: MFMA part */         \
            RAW_BARRIER();                                                                                     \
        }                                                                                                      \
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RAW_BARRIER();                      // weight stages 0, 1 and sub-patch 0 have landed
    if (g == 1) RAW_BARRIER();          // stagger
    half8 af[TI], bf[4];
    for (int chunk = 0; chunk < nC; ++chunk) {
        const bool last = chunk + 1 == nC;
        const int c32 = chunk * 32;
        const int nb = (chunk + 1) & 1;
        const char* pb = pbuf + (chunk & 1) * G::PBUF;
        PW_STEP(0) PW_STEP(1) PW_STEP(2) PW_STEP(3) PW_STEP(4) PW_STEP(5) PW_STEP(6) PW_STEP(7) PW_STEP(8)
    }
    if (g == 0) RAW_BARRIER();          // re-align the two groups before the epilogue reuses the LDS
#undef PW_STEP
#undef END_OF_STEP_WAIT
#undef WAIT_VM
#undef ISSUE_W
#undef ISSUE_W_HALF
#undef ISSUE_P
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- fused 1x1 strided shortcut (BasicBlock downsample path, resnet18.py:42-45): Cin2 / 32 more K-steps on the block
    //      input sampled at the output pixels' positions.  All eight waves in lockstep, two [weights | pixels] stages
    //      (32 or 40 KB) in the LDS the main loop has left (every DMA is drained, every fragment read retired), one barrier per
    //      step: the next step's tiles fly while this one computes.  4-8 steps against 72-144 of the main loop. ----
    if (a.in2) {
        const int nC2 = a.Cin2 / 32;
        constexpr int XROWS = G::PX / 128, SST = G::WST + G::PX * 64;   // pixel pieces per thread; bytes of one stage
        const _Float16* w2src[G::WROWS];
        int x2off[XROWS];                                       // 31-bit element offsets (checked by the launcher), -1: zero page
        const int lg = ((tid & 3) ^ (((tid >> 4) & 1) << 1)) * 8;   // logical chunk held at position tid & 3 ((row >> 2) & 1 == (tid >> 4) & 1)
#pragma unroll
        for (int i = 0; i < G::WROWS; ++i) w2src[i] = a.wgt2 + (size_t)(ch0 + (tid >> 2) + 128 * i) * a.Cin2 + lg;
#pragma unroll
        for (int i = 0; i < XROWS; ++i) {
            const int row = (tid >> 2) + 128 * i;               // tile pixel of this thread's piece
            const int n = n0 + G::p_img(row);
            x2off[i] = n < a.N ? (int)((((size_t)(map_image<IMAP>(a, n) % a.in2_mod) * a.H2 + (size_t)G::p_oy(row) * a.stride2) * a.W2 +
                                        (size_t)G::p_ox(row) * a.stride2) * a.Cin2 + lg)
                               : -1;
        }
#define ISSUE_S(C2, ST)                                                                                        \
    {                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < G::WROWS; ++i)                                                   \
            GLDS16(w2src[i] + (C2) * 32, smem + (ST) * SST + (i * 512 + wave * 64) * 16);                      \
        _Pragma("unroll") for (int i = 0; i < XROWS; ++i)                                                      \
            GLDS16(x2off[i] >= 0 ? a.in2 + (size_t)(unsigned)x2off[i] + (C2) * 32 : (const _Float16*)g_zero_page_pw, \
                   smem + (ST) * SST + G::WST + (i * 512 + wave * 64) * 16);                                   \
    }
        RAW_BARRIER();                   // both groups are past their last fragment reads
        ISSUE_S(0, 0);
        for (int c2 = 0; c2 < nC2; ++c2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            RAW_BARRIER();               // stage c2 & 1 has landed for every wave; everyone is done with the other stage
            if (c2 + 1 < nC2) ISSUE_S(c2 + 1, (c2 + 1) & 1);
            const char* ss = smem + (c2 & 1) * SST;
            half8 sa[TI], sb[TP];
#pragma unroll
            for (int i = 0; i < TI; ++i) sa[i] = *(const half8*)(ss + a_off + i * 16 * 64 + a_byte);
#pragma unroll
            for (int j = 0; j < TP; ++j) sb[j] = *(const half8*)(ss + G::WST + (pbase + 16 * j + l16) * 64 + a_byte);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) acc[i][j] = mfma_16x16x32<BF>(sa[i], sb[j], acc[i][j]);
        }
#undef ISSUE_S
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RAW_BARRIER();                   // the epilogue reuses the stages
    }
#undef RAW_BARRIER

    // ---- epilogue: each channel half (4 waves) in its own 64 KB (conv_epilogue.h) ----
    const int chg = ch0 + 128 * g;
    auto pixmap = [&](int p, int& n, int& rem) -> bool {
        n = n0 + G::p_img(p);
        rem = G::p_oy(p) * TW + G::p_ox(p);
        const bool ok = n < a.N;
        n = map_image<IMAP>(a, n);
        return ok;
    };
    auto offmap = [&](int p, size_t& off) -> bool {
        int n, rem;
        const bool ok = pixmap(p, n, rem);
        off = ((size_t)n * (TH * TW) + rem) * a.Cout;
        return ok;
    };
    if constexpr (POOL) epilogue_lite<TJ, BF, false, true>(a, acc, smem + g * BMI_EPILOGUE_LDS_BYTES, tid & 255, chg, pixmap, offmap);
    else epilogue_coalesced<TJ, EPI, 16, BF>(a, acc, smem + g * BMI_EPILOGUE_LDS_BYTES, tid & 255, chg, pixmap, offmap);
}

// ---------------------------------------------------------------------------------------------------------------------------
// conv3x3_pw4: the same tile, LDS layout, DMA sources, K order and epilogue code with FOUR waves — one per SIMD, wave tile 128 ch x
// 128 px (8 x 8 tiles of v_mfma_f32_16x16x32: 256 accumulator registers, the unified 512-entry file of a one-wave-per-SIMD
// workgroup) — and NO hand-over of the matrix pipe between wave groups: a wave reads the fragments of K-step s + 1 into a second
// register set while its 64 MFMAs of step s run (software pipelining inside one instruction stream), 16 ds_read_b128 per 64 MFMAs
// instead of 12 per 32, ONE barrier per K-step instead of four.  Wave (wc, wp) owns the channels of conv3x3_pw's waves (0, wc, wp)
// and (1, wc, wp): rows wc*64 + 16 i of both 128-channel halves, so each half is finished by conv_epilogue.h exactly as there.
//   step s:  wait (own DMA of step s + 1 landed) -> barrier (everyone's landed; everyone's fragment reads of step s are done, so
//            stage s % 3 is free) -> DMA of step s + 3 into stage s % 3 (+ two pieces of the next sub-patch) -> 16 fragment reads
//            of step s + 1 from stage (s + 1) % 3 into the other set, interleaved with the 64 MFMAs of step s.
// Two chunks (18 K-steps) per loop iteration: the register set alternates per step and 9 is odd (Cin % 64 == 0).
template <int TW, int EPI, bool BF, bool IMAP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_pw4_kernel(ConvArgs a) {
    using G = PwGeom<TW>;
    constexpr int CT = G::CT, TH = G::TH, IMGS = G::IMGS, PH = G::PH, PW = G::PW, PWP = G::PWP;
    constexpr int TP = 8, NPC = 2 * G::ITER_P;                       // pixel tiles per wave; patch pieces per thread and sub-patch
    typedef float accv __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) char smem[G::LDS_BYTES];
    char* const wst = smem;
    char* const pbuf = smem + G::NST * G::WST;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, kq = lane >> 4;
    const int wc = wave >> 1, wp = wave & 1;

    const int n_ctiles = a.Cout / CT;
    const int n_ptiles = (a.N + IMGS - 1) / IMGS;
    int ptile, ctile;
    xcd_tile_map(blockIdx.x, n_ptiles, n_ctiles, ptile, ctile, a.xcd_split);
    const int ch0 = ctile * CT;
    const int n0 = ptile * IMGS;
    const int Ktot = 9 * a.Cin;
    const int nC = a.Cin / 32;

    // weights: piece q = tid + 256 i -> row (tid >> 2) + 64 i, position tid & 3 holds chunk pos ^ 2 ((row >> 2) & 1) ((row >> 2) & 1 == (tid >> 4) & 1)
    const _Float16* const wsrc0 = a.wgt + (size_t)(ch0 + (tid >> 2)) * Ktot + ((tid & 3) ^ (((tid >> 4) & 1) << 1)) * 8;
    const size_t wrow64 = (size_t)64 * Ktot;
#define ISSUE_W4(KOFF, ST)                                                                       \
    {                                                                                            \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                            \
            GLDS16(wsrc0 + i * wrow64 + (KOFF), wst + (ST) * G::WST + (i * 256 + wave * 64) * 16); \
    }
    ISSUE_W4(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    int psrc[NPC];
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
        const int q = tid + 256 * i;
        const int cell = q >> 2, pos = q & 3;
        const int rowc = cell / PWP, x = cell - rowc * PWP;
        const int img = rowc / PH, y = rowc - img * PH;
        const int n = n0 + img;
        const int iy = y - 1, ix = x - 1;
        const bool ok = cell < G::CELLS && x < PW && n < a.N && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        psrc[i] = ok ? (int)((((size_t)(map_image<IMAP>(a, n) % a.in_mod) * a.H + iy) * a.W + ix) * a.Cin + (pos ^ ((y & 1) << 1)) * 8) : -1;
    }
#define ISSUE_P4(I, C0, PB)                                                                                  \
    GLDS16(psrc[I] >= 0 ? a.in + (size_t)(unsigned)psrc[I] + (C0) : (const _Float16*)g_zero_page_pw,       \
           pbuf + (PB) * G::PBUF + ((I) * 256 + wave * 64) * 16)
#pragma unroll
    for (int i = 0; i < NPC; ++i) ISSUE_P4(i, 0, 0);
    ISSUE_W4(a.Cin, 1);                                        // K-step 1 = tap 1 of chunk 0
    ISSUE_W4(2 * a.Cin, 2);                                    // K-step 2
    __builtin_amdgcn_sched_barrier(0);

    // fragment geometry: A rows h*128 + wc*64 + 16 i + l16 (I = 4 h + i), B as conv3x3_pw
    const int a_off = (wc * 64 + l16) * 64 + ((kq ^ (((l16 >> 2) & 1) << 1)) << 4);
    const int pbase = wp * 128;
    const int wave_cell = (G::p_img(pbase) * PH + G::p_oy(pbase)) * PWP + G::p_ox(pbase);
    int boff[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
        boff[ky] = (wave_cell + (l16 >> 2) * PWP + (l16 & 3)) * 64 + ((kq ^ ((((l16 >> 2) + ky) & 1) << 1)) << 4);

    accv acc[8][TP];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

#define WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define BARRIER4()                                     \
    {                                                  \
        asm volatile("" ::: "memory");                 \
        __builtin_amdgcn_s_barrier();                  \
        asm volatile("" ::: "memory");                 \
    }
#define SB() __builtin_amdgcn_sched_barrier(0)
    // The accumulators live in AGPRs and every MFMA updates its tile IN PLACE (inline asm, "+a"): with the builtin hipcc gave the
    // 256 accumulator registers different source and destination tuples and copied them through VGPRs around every MFMA.
#define MFMA4(ACC, A, B)                                                                                       \
    {                                                                                                          \
        if constexpr (BF) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B)); \
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B));              \
    }
    // fragment reads of K-step (tap TAP, sub-patch at LDS offset PBO): channel rows h*128 + wc*64 + 16 i (AF = half h), pixel tiles
    // 0..7.  Inline asm with hand-counted lgkmcnt waits: in front of an asm MFMA hipcc waits for EVERY LDS read in flight, including
    // the four of the other channel half issued a moment ago (a full LDS round trip per K-step in the first build).
#define DSR(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define READ_A(TAP, H, AF)                                                                                     \
    {                                                                                                          \
        DSR(AF[0], a_lds, ((TAP) % 3) * G::WST + (H) * 8192 + 0 * 1024);                                       \
        DSR(AF[1], a_lds, ((TAP) % 3) * G::WST + (H) * 8192 + 1 * 1024);                                       \
        DSR(AF[2], a_lds, ((TAP) % 3) * G::WST + (H) * 8192 + 2 * 1024);                                       \
        DSR(AF[3], a_lds, ((TAP) % 3) * G::WST + (H) * 8192 + 3 * 1024);                                       \
    }
#define READ_B(TAP, PBO, BFR)                                                                                  \
    {                                                                                                          \
        constexpr int ky_ = (TAP) / 3, kx_ = (TAP) - 3 * ky_;                                                  \
        const unsigned pb_ = b_lds[ky_] + (PBO);                                                               \
        DSR(BFR[0], pb_, (ky_ * PWP + kx_) * 64 + G::cell_delta(0) * 64);                                      \
        DSR(BFR[1], pb_, (ky_ * PWP + kx_) * 64 + G::cell_delta(1) * 64);                                      \
        DSR(BFR[2], pb_, (ky_ * PWP + kx_) * 64 + G::cell_delta(2) * 64);                                      \
        DSR(BFR[3], pb_, (ky_ * PWP + kx_) * 64 + G::cell_delta(3) * 64);                                      \
        DSR(BFR[4], pb_, (ky_ * PWP + kx_) * 64 + G::cell_delta(4) * 64);                                      \
        DSR(BFR[5], pb_, (ky_ * PWP + kx_) * 64 + G::cell_delta(5) * 64);                                      \
        DSR(BFR[6], pb_, (ky_ * PWP + kx_) * 64 + G::cell_delta(6) * 64);                                      \
        DSR(BFR[7], pb_, (ky_ * PWP + kx_) * 64 + G::cell_delta(7) * 64);                                      \
    }
    // What may be in flight when step TAP takes its barrier (before its own DMA is issued): the DMA instructions of the step before — 4
    // weight pieces (unless that step had no step + 3 inside the tile) and 2 sub-patch pieces (taps 1 .. ITER_P of a chunk with a successor)
#define TOP_WAIT(TAP)                                                                                          \
    {                                                                                                          \
        constexpr int tp_ = ((TAP) + 8) % 9;                       /* tap of the previous step */              \
        constexpr int pp_ = (tp_ >= 1 && tp_ <= G::ITER_P) ? 2 : 0;                                            \
        const bool plast_ = (TAP) == 0 ? false : last;             /* (tap 8 of the previous chunk: never a last one) */ \
        if (!plast_) { WAIT_VM(4 + pp_); }                                                                     \
        else if (tp_ <= 5) { WAIT_VM(4); }                                                                     \
        else { WAIT_VM(0); }                                                                                   \
    }
    // One K-step.  Entering: af_lo / af_hi = this step's channel fragments (the af_hi reads may still be in flight), BX = its pixel
    // fragments.  Leaving: af_lo / af_hi / BY hold the next step's.
#define STEP4(TAP, BX, BY)                                                                                     \
    {                                                                                                          \
        constexpr int tn_ = (TAP) == 8 ? 0 : (TAP) + 1;                                                        \
        const unsigned pn_ = (TAP) == 8 ? pbn : pb;                /* sub-patch of the next step (LDS offset) */ \
        const bool more_ = (TAP) < 8 || !last;                     /* a next step exists */                    \
        /* LDS reads return in order: pixel fragments (8), af_lo (4), af_hi (4) — all but af_hi have landed */ \
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");                                                     \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) MFMA4(acc[i][j], af_lo[i], BX[j]);                   \
        SB();                                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         /* this wave has read stage TAP % 3 completely */ \
        TOP_WAIT(TAP);                                             /* its DMA of the next step's stage has landed */ \
        BARRIER4();                                                                                            \
        if ((TAP) < 6) { ISSUE_W4(((TAP) + 3) * a.Cin + c32, (TAP) % 3); }                                     \
        else if (!last) { ISSUE_W4(((TAP) - 6) * a.Cin + c32 + 32, (TAP) % 3); }                               \
        if ((TAP) >= 1 && (TAP) <= G::ITER_P && !last) {                                                       \
            ISSUE_P4(2 * ((TAP) - 1), c32 + 32, nb);                                                           \
            ISSUE_P4(2 * ((TAP) - 1) + 1, c32 + 32, nb);                                                       \
        }                                                                                                      \
        if (more_) { READ_B(tn_, pn_, BY); }                                                                   \
        SB();                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
            _Pragma("unroll") for (int j = 2; j < TP; ++j) MFMA4(acc[i][j], af_lo[i], BX[j]);                  \
        SB();                                                                                                  \
        if (more_) { READ_A(tn_, 0, af_lo); }                                                                  \
        SB();                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
            _Pragma("unroll") for (int j = 0; j < TP; ++j) MFMA4(acc[4 + i][j], af_hi[i], BX[j]);              \
        SB();                                                                                                  \
        if (more_) { READ_A(tn_, 1, af_hi); }                                                                  \
        SB();                                                                                                  \
    }

    WAIT_VM(0);
    BARRIER4();                          // stages 0, 1, 2 and sub-patch 0 have landed
    half8 af_lo[4], af_hi[4], bf0[TP], bf1[TP];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned a_lds = lds0 + a_off;
    unsigned b_lds[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) b_lds[ky] = lds0 + G::NST * G::WST + boff[ky];
    READ_B(0, 0u, bf0);
    READ_A(0, 0, af_lo);
    READ_A(0, 1, af_hi);
    for (int chunk = 0; chunk < nC; chunk += 2) {
        {
            const bool last = false;     // (nC is even: an even chunk always has a successor)
            const int c32 = chunk * 32;
            const int nb = 1;
            const unsigned pb = 0u, pbn = G::PBUF;
            STEP4(0, bf0, bf1) STEP4(1, bf1, bf0) STEP4(2, bf0, bf1) STEP4(3, bf1, bf0) STEP4(4, bf0, bf1)
            STEP4(5, bf1, bf0) STEP4(6, bf0, bf1) STEP4(7, bf1, bf0) STEP4(8, bf0, bf1)
        }
        {
            const bool last = chunk + 2 == nC;
            const int c32 = (chunk + 1) * 32;
            const int nb = 0;
            const unsigned pb = G::PBUF, pbn = 0u;
            STEP4(0, bf1, bf0) STEP4(1, bf0, bf1) STEP4(2, bf1, bf0) STEP4(3, bf0, bf1) STEP4(4, bf1, bf0)
            STEP4(5, bf0, bf1) STEP4(6, bf1, bf0) STEP4(7, bf0, bf1) STEP4(8, bf1, bf0)
        }
    }
#undef STEP4
#undef TOP_WAIT
#undef READ_A
#undef READ_B
#undef DSR
#undef MFMA4
#undef SB
#undef ISSUE_P4
#undef ISSUE_W4
#undef WAIT_VM
    // (the asm MFMAs are opaque to hipcc's hazard recogniser: their results are read — v_accvgpr_read — only behind these wait states)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    BARRIER4();                          // every wave is done with the stages: the epilogue reuses the LDS
#undef BARRIER4

    // ---- epilogue: the two 128-channel halves one after the other, each exactly as a wave group of conv3x3_pw finishes it ----
    auto pixmap = [&](int p, int& n, int& rem) -> bool {
        n = n0 + G::p_img(p);
        rem = G::p_oy(p) * TW + G::p_ox(p);
        const bool ok = n < a.N;
        n = map_image<IMAP>(a, n);
        return ok;
    };
    auto offmap = [&](int p, size_t& off) -> bool {
        int n, rem;
        const bool ok = pixmap(p, n, rem);
        off = ((size_t)n * (TH * TW) + rem) * a.Cout;
        return ok;
    };
    typedef accv acc_half[4][TP];
    epilogue_coalesced<4, EPI, 16, BF>(a, *(acc_half*)&acc[0], smem, tid, ch0, pixmap, offmap);
    epilogue_coalesced<4, EPI, 16, BF>(a, *(acc_half*)&acc[4], smem, tid, ch0 + 128, pixmap, offmap);
}

// ---------------------------------------------------------------------------------------------------------------------------
// conv3x3_pwp (round 4): conv3x3_pw made PERSISTENT for the launches that finish in the plain epilogue (BN + ReLU, with or without the fused
// shortcut).  One workgroup per CU walks the tiles and the main loop simply CONTINUES across them: in the last chunk of a tile the K-steps
// that would prefetch "the next chunk" — the weight stages of steps 0 / 1 (issued at taps 7 / 8) and the pieces of sub-patch 0 (taps
// 1..ITER_P) — fetch the NEXT TILE's instead, with the unchanged counted-vmcnt schedule; Cin % 64 == 0 makes the last chunk odd, so what it
// prefetches lands in stages 0 / 1 and sub-patch buffer 0.  The LDS map [W0 | W1 | P0 | W2 | P1 | pad] keeps those three outside the
// 64 KB [W2 | P1 | pad] the shortcut's K-steps work in (and the epilogue did before PWP_DIRECT: conv3x3_s2's two rounds of 128 pixels per channel
// half; BN comes from an LDS table — a global load there would wait for every DMA in flight).  A full tile leaves its 16 output stores per thread in flight
// across the tile boundary (vmcnt(16)).  Why: per-tile fixed cost of conv3x3_pw from its own K = 2304 / 4608 rates (1308 / 1422 TFLOP/s, corrected
// for the last partial round of tiles): 14-21 K-steps' worth per tile, of which the prologue's HBM round trip and the workgroup hand-over are
// what a persistent walk removes.  Same K order, same arithmetic, same bits as conv3x3_pw_kernel<TW, PLAIN>.
template <int TW>
struct PwpGeom {
    using G = PwGeom<TW>;
    static constexpr int WST = G::WST, PBUF = G::PBUF;
    __host__ __device__ static constexpr int w_off(int st) { return st < 2 ? st * WST : 2 * WST + PBUF; }
    __host__ __device__ static constexpr int p_off(int b) { return b == 0 ? 2 * WST : 3 * WST + PBUF; }
    static constexpr int E_OFF = 2 * WST + PBUF;                                        // [W2 | P1 | pad]
    static constexpr int MAIN_END = 3 * WST + 2 * PBUF;
    static constexpr int E_END = E_OFF + 65536 > MAIN_END ? E_OFF + 65536 : MAIN_END;
    static constexpr int BN_OFF = E_END;                                                // fp32 scale[512] | bias[512]
    static constexpr int LDS_BYTES = BN_OFF + 4096;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

#ifndef PWP_DIRECT
#define PWP_DIRECT 1      // the MFMA rows of a wave's four channel tiles are a PERMUTATION of its 64 channels — row r of tile i is channel
                          // 32 (i >> 1) + 8 (r >> 2) + 4 (i & 1) + (r & 3) — so that a lane's accumulators are, per pixel, two runs of 8 consecutive
                          // channels: the epilogue stores them straight from the registers (two 16-byte stores per pixel tile; the four lanes of a
                          // pixel write 64 contiguous bytes per store) with no trip through LDS and no barrier.  0: conv3x3_s2's epilogue through LDS
#endif
// EPIK: BMI_EPI_PLAIN, or (PWP_DIRECT only) BMI_EPI_LITE_RES / BMI_EPI_LITE_RES_MC — the BasicBlock tails (residual whose rows are the output's rows,
// ReLU, optionally the 2-bit elementwise site) finished straight from the registers too: a lane fetches the two 16-byte residual runs of each of its
// pixel tiles itself (the four lanes of a pixel read 64 contiguous bytes per instruction), so the lite epilogue's residual DMA, its 2 x 64 KB of LDS and
// its three barriers are gone and these launches can take the persistent walk.  Same arithmetic in the same order as epilogue_lite: the same bits.
// POOLP (4x4 maps, ConvArgs::pool): the tail feeds nothing but an exit head — instead of the map, fp32 means over it ([row][Cout]) of relu(.): a pixel
// tile is one image, its 16 pixels the 16 lanes of a DPP row (epilogue_lite's POOL: the same four adds in the same order).
template <int TW, bool BF, bool SHORTCUT, int EPIK = BMI_EPI_PLAIN, bool POOLP = false>
__global__ __launch_bounds__(512, 1) void conv3x3_pwp_kernel(ConvArgs a, int n_tiles) {
    static_assert(!POOLP || (TW == 4 && EPIK != BMI_EPI_PLAIN), "pooled output: the 4x4 maps' BasicBlock tails");
    static_assert(EPIK == BMI_EPI_PLAIN || (PWP_DIRECT && !SHORTCUT && (EPIK == BMI_EPI_LITE_RES || EPIK == BMI_EPI_LITE_RES_MC || EPIK == BMI_EPI_LITE_RES_MSK)),
                  "epilogue kind");
    using G = PwGeom<TW>;
    using L = PwpGeom<TW>;
    constexpr int CT = G::CT, TH = G::TH, IMGS = G::IMGS, PH = G::PH, PW = G::PW, PWP = G::PWP;
    constexpr int TI = 4, TP = 8;
    typedef float accv __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) char smem[L::LDS_BYTES];
    float* const bn_scale = (float*)(smem + L::BN_OFF);
    float* const bn_bias = bn_scale + 512;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, kq = lane >> 4;
    const int g = wave >> 2, wc = (wave >> 1) & 1, wp = wave & 1;
    const int n_ctiles = a.Cout / CT;
    const int n_ptiles = (a.N + IMGS - 1) / IMGS;
    const int Ktot = 9 * a.Cin;
    const int nC = a.Cin / 32;                                   // even (launcher: Cin % 64 == 0)

    for (int c = tid; c < a.Cout; c += 512) {
        bn_scale[c] = (a.scale ? a.scale[c] : 1.f) * a.out_mul;
        bn_bias[c] = a.bias ? a.bias[c] : 0.f;
    }
    __syncthreads();   // (no LDS-DMA in flight yet)

    // Every DMA is a buffer_load ... lds through a per-tile buffer descriptor (wave-uniform, SGPRs) with a 32-bit per-lane byte offset that
    // is computed ONCE per kernel (conv3x3_s2's scheme: with per-tile 64-bit sources the persistent loop spilled 30-60 VGPRs): what a piece
    // reads is the same for every tile up to the tile's first image, and a lane whose offset lies beyond the descriptor loads ZEROS — the
    // padding ring, the cells of the pitch, the images of a last tile beyond N.
    // weights: piece q = tid + 512 i -> row (tid >> 2) + 128 i of the channel tile, position tid & 3 holds chunk pos ^ 2 ((row >> 2) & 1)
    unsigned woff[G::WROWS];
#pragma unroll
    for (int i = 0; i < G::WROWS; ++i) {
        const int row = (tid >> 2) + 128 * i;
        // (PWP_DIRECT: a fragment's 16 lanes read rows 8 a + b + const, a = 0..3, b = 0..3: the chunk position alternates with row >> 3)
        woff[i] = 2u * ((unsigned)row * Ktot + (((tid & 3) ^ (((row >> (PWP_DIRECT ? 3 : 2)) & 1) << 1)) << 3));
    }
    const unsigned wbytes = 2u * (unsigned)CT * Ktot;
    const unsigned HWC = (unsigned)a.H * a.W * a.Cin;
    constexpr unsigned OOB = 0xfffffff0u;
    // sub-patch: piece q = tid + 512 i -> cell q >> 2, position q & 3 holds chunk pos ^ 2 (y & 1): byte offset relative to the tile's image 0
    unsigned pre[G::ITER_P];
#pragma unroll
    for (int i = 0; i < G::ITER_P; ++i) {
        const int q = tid + 512 * i;
        const int cell = q >> 2, pos = q & 3;
        const int rowc = cell / PWP, x = cell - rowc * PWP;
        const int img = rowc / PH, y = rowc - img * PH;
        const int iy = y - 1, ix = x - 1;
        const bool ok = cell < G::CELLS && x < PW && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        pre[i] = ok ? 2u * ((unsigned)img * HWC + (unsigned)((iy * a.W + ix) * a.Cin + ((pos ^ ((y & 1) << 1)) << 3))) : OOB;
    }
    __amdgpu_buffer_rsrc_t rs_w, rs_wn, rs_in;
#define BLDS16(RSRC, VOFF, SOFF, LDSPTR) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds((RSRC), (__attribute__((address_space(3))) void*)(LDSPTR), 16, (int)(VOFF), (int)(SOFF), 0, 0)
#define ISSUE_W(RS, KOFF, ST)                                                                      \
    {                                                                                              \
        const unsigned so_ = __builtin_amdgcn_readfirstlane(2u * (unsigned)(KOFF));                \
        _Pragma("unroll") for (int i = 0; i < G::WROWS; ++i)                                       \
            BLDS16(RS, woff[i], so_, smem + L::w_off(ST) + (i * 512 + wave * 64) * 16);            \
    }
#define ISSUE_P(I, C0, PB) \
    BLDS16(rs_in, pre[I], __builtin_amdgcn_readfirstlane(2u * (unsigned)(C0)), smem + L::p_off(PB) + ((I) * 512 + wave * 64) * 16)
#define TILE_RSRC_W(RS, CH0) RS = __builtin_amdgcn_make_buffer_rsrc((void*)(a.wgt + (size_t)(CH0) * Ktot), 0, wbytes, 0x00020000)
#define TILE_RSRC_IN(N0)                                                                                                      \
    {                                                                                                                         \
        const int nimg_ = a.N - (N0) < IMGS ? a.N - (N0) : IMGS;                                                              \
        rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)(N0) * HWC), 0, 2u * (unsigned)nimg_ * HWC, 0x00020000); \
    }

    // ---- per-lane fragment geometry (tile-independent) ----
    const int a_off = (g * 128 + wc * 64 + (PWP_DIRECT ? 8 * (l16 >> 2) + (l16 & 3) : l16)) * 64;
    // byte offset of channel tile i's row of this lane relative to a_off
#define A_TILE(I) (PWP_DIRECT ? (32 * ((I) >> 1) + 4 * ((I) & 1)) * 64 : (I) * 16 * 64)
    const int pbase = wp * 128;
    const int a_byte = (kq ^ (((l16 >> 2) & 1) << 1)) << 4;
    const int wave_cell = (G::p_img(pbase) * PH + G::p_oy(pbase)) * PWP + G::p_ox(pbase);
    int boff[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
        boff[ky] = (wave_cell + (l16 >> 2) * PWP + (l16 & 3)) * 64 + ((kq ^ ((((l16 >> 2) + ky) & 1) << 1)) << 4);

#define RAW_BARRIER()                                  \
    {                                                  \
        __builtin_amdgcn_sched_barrier(0);             \
        asm volatile("" ::: "memory");                 \
        __builtin_amdgcn_s_barrier();                  \
        asm volatile("" ::: "memory");                 \
        __builtin_amdgcn_sched_barrier(0);             \
    }
#define WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
    // (step 0 of a tile that follows a full tile: what step 1 reads — weight stage 1 — was issued BEFORE the previous tile's 16 output stores and
    //  has landed at the tile's start; only this step's own weight DMA is younger than the stores, so the stores may stay in flight one step longer)
#define END_OF_STEP_WAIT(TAP)                                                                                  \
    {                                                                                                          \
        constexpr int pieces_ = ((TAP) >= 1 && (TAP) <= G::ITER_P ? 1 : 0) + ((TAP) >= 2 && (TAP) <= G::ITER_P + 1 ? 1 : 0); \
        if ((TAP) == 0 && chunk == 0 && stores16) { WAIT_VM(16 + G::WROWS); }                                  \
        else if (!last) { WAIT_VM(G::WROWS + pieces_); }                                                       \
        else if ((TAP) >= 7) { WAIT_VM(0); }                                                                   \
        else { WAIT_VM(G::WROWS); }                                                                            \
    }
    // One K-step (conv3x3_pw's PW_STEP): what it prefetches for "the next chunk" comes from (nx_w, nx_k, nx_p): this tile's next chunk, or
    // chunk 0 of the next tile
#define PWP_STEP(TAP)                                                                                          \
    {                                                                                                          \
        constexpr int ky_ = (TAP) / 3, kx_ = (TAP) - 3 * ky_;                                                  \
        const char* ws_ = smem + L::w_off((TAP) % 3) + a_off + a_byte;                                         \
        const char* pb_ = pb + (ky_ * PWP + kx_) * 64;                                                         \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                     \
            if (kk == 0) {                                                                                     \
                _Pragma("unroll") for (int i = 0; i < TI; ++i) af[i] = *(const half8*)(ws_ + A_TILE(i));       \
            }                                                                                                  \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) bf[j] = *(const half8*)(pb_ + boff[ky_] + G::cell_delta(4 * kk + j) * 64); \
            if (kk == 0) {                                                                                     \
                if ((TAP) < 7) { ISSUE_W(rs_w, ((TAP) + 2) * a.Cin + c32, ((TAP) + 2) % 3); }                  \
                else if (!last) {                                                                              \
                    if (tile_last) { ISSUE_W(rs_wn, ((TAP) - 7) * a.Cin, ((TAP) + 2) % 3); }                   \
                    else { ISSUE_W(rs_w, ((TAP) - 7) * a.Cin + c32 + 32, ((TAP) + 2) % 3); }                   \
                }                                                                                              \
                if ((TAP) >= 1 && (TAP) <= G::ITER_P && !last) {                                               \
                    constexpr int i_ = (TAP) >= 1 && (TAP) <= G::ITER_P ? (TAP) - 1 : 0;                       \
                    ISSUE_P(i_, nx_k, nb);                                                                     \
                }                                                                                              \
            }                                                                                                  \
            if (kk == 1 && g == 1) END_OF_STEP_WAIT(TAP);                                                      \
            RAW_BARRIER();                                                                                     \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
            __builtin_amdgcn_s_setprio(1);                                                                     \
            _Pragma("unroll") for (int i = 0; i < TI; ++i)                                                     \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                  \
                    acc[i][4 * kk + j] = mfma_16x16x32<BF>(af[i], bf[j], acc[i][4 * kk + j]);                  \
            __builtin_amdgcn_s_setprio(0);                                                                     \
            if (kk == 1 && g == 0) END_OF_STEP_WAIT(TAP);                                                      \
            RAW_BARRIER();                                                                                     \
        }                                                                                                      \
    }

    int vb = blockIdx.x;
    int ptile, ctile;
    xcd_tile_map(vb, n_ptiles, n_ctiles, ptile, ctile, a.xcd_split);
    int ch0 = ctile * CT, n0 = ptile * IMGS;
    TILE_RSRC_W(rs_w, ch0);
    TILE_RSRC_W(rs_wn, ch0);          // (descriptors are rebuilt, never copied: the opaque type has no host-side copy and the host pass drops the kernel)
    TILE_RSRC_IN(n0)
    ISSUE_W(rs_w, 0, 0);
#pragma unroll
    for (int i_ = 0; i_ < G::ITER_P; ++i_) { ISSUE_P(i_, 0, 0); }
    ISSUE_W(rs_w, a.Cin, 1);
    __builtin_amdgcn_sched_barrier(0);
    bool stores16 = false;
    while (true) {
        const int nvb = vb + (int)gridDim.x;
        const bool more = nvb < n_tiles;
        int ch0n = 0, n0n = 0;
        if (more) {
            int pt, ct;
            xcd_tile_map(nvb, n_ptiles, n_ctiles, pt, ct, a.xcd_split);
            ch0n = ct * CT; n0n = pt * IMGS;
            TILE_RSRC_W(rs_wn, ch0n);
        }
        accv acc[TI][TP];
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

        // weight stages 0, 1 and sub-patch 0 have landed; the previous tile's 16 output stores per thread were issued behind them
        if (!stores16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        RAW_BARRIER();
        if (g == 1) RAW_BARRIER();          // stagger
        half8 af[TI], bf[4];
        for (int chunk = 0; chunk < nC; ++chunk) {
            const bool tile_last = chunk + 1 == nC;
            const bool last = tile_last && !more;
            const int c32 = chunk * 32;
            const int nb = (chunk + 1) & 1;
            const char* pb = smem + L::p_off(chunk & 1);
            const int nx_k = tile_last ? 0 : c32 + 32;
            // the last chunk of a tile issues no piece of its own tile any more: the input descriptor becomes the next tile's here
            if (tile_last && more) { TILE_RSRC_IN(n0n) }
            PWP_STEP(0) PWP_STEP(1) PWP_STEP(2) PWP_STEP(3) PWP_STEP(4) PWP_STEP(5) PWP_STEP(6) PWP_STEP(7) PWP_STEP(8)
        }
        if (g == 0) RAW_BARRIER();          // re-align the two groups: every wave is past its last fragment reads

        if constexpr (SHORTCUT) {
            // fused 1x1 strided shortcut (conv3x3_pw's: Cin2 / 32 lock-step K-steps, two [weights | pixels] stages) in the epilogue's 64 KB
            const int nC2 = a.Cin2 / 32;
            constexpr int XROWS = G::PX / 128, SST = G::WST + G::PX * 64;
            char* const sbase = smem + L::E_OFF;
            const int lg = ((tid & 3) ^ (((tid >> 4) & 1) << 1)) * 8;                                  // pixel rows: chunk position alternates with row >> 2
            const int lgw = PWP_DIRECT ? ((tid & 3) ^ (((tid >> 5) & 1) << 1)) * 8 : lg;               // weight rows (PWP_DIRECT): with row >> 3
            int x2off[XROWS];
#pragma unroll
            for (int i = 0; i < XROWS; ++i) {
                const int row = (tid >> 2) + 128 * i;
                const int n = n0 + G::p_img(row);
                x2off[i] = n < a.N ? (int)((((size_t)(n % a.in2_mod) * a.H2 + (size_t)G::p_oy(row) * a.stride2) * a.W2 +
                                            (size_t)G::p_ox(row) * a.stride2) * a.Cin2 + lg)
                                   : -1;
            }
#define ISSUE_S(C2, ST)                                                                                        \
    {                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < G::WROWS; ++i)                                                   \
            GLDS16(a.wgt2 + (size_t)(ch0 + (tid >> 2) + 128 * i) * a.Cin2 + lgw + (C2) * 32, sbase + (ST) * SST + (i * 512 + wave * 64) * 16); \
        _Pragma("unroll") for (int i = 0; i < XROWS; ++i)                                                      \
            GLDS16(x2off[i] >= 0 ? a.in2 + (size_t)(unsigned)x2off[i] + (C2) * 32 : (const _Float16*)g_zero_page_pw, \
                   sbase + (ST) * SST + G::WST + (i * 512 + wave * 64) * 16);                                  \
    }
            ISSUE_S(0, 0);
            for (int c2 = 0; c2 < nC2; ++c2) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                RAW_BARRIER();
                if (c2 + 1 < nC2) ISSUE_S(c2 + 1, (c2 + 1) & 1);
                const char* ss = sbase + (c2 & 1) * SST;
                half8 sa[TI], sb[TP];
#pragma unroll
                for (int i = 0; i < TI; ++i) sa[i] = *(const half8*)(ss + a_off + A_TILE(i) + a_byte);
#pragma unroll
                for (int j = 0; j < TP; ++j) sb[j] = *(const half8*)(ss + G::WST + (pbase + 16 * j + l16) * 64 + a_byte);
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TP; ++j) acc[i][j] = mfma_16x16x32<BF>(sa[i], sb[j], acc[i][j]);
            }
#undef ISSUE_S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            RAW_BARRIER();
        }

        if constexpr (EPIK != BMI_EPI_PLAIN) {
            // ---- BasicBlock tail straight from the registers (see EPIK above) ----
            constexpr bool MC = EPIK == BMI_EPI_LITE_RES_MC, MSKS = EPIK == BMI_EPI_LITE_RES_MSK;
            const int chw = ch0 + 128 * g + wc * 64 + 8 * kq;
            // the residual runs of this lane: run h of pixel tile j = 8 channels at chw + 32 h (rows beyond the tensor read row 0, never stored)
            half8_e rres[2][TP];
            unsigned poff[TP];                                    // element offset of the pixel's row / 8 (< 2^32: launcher)
            int mrow[MSKS ? TP : 1];                              // Masksembles: element offset of this lane's multipliers in the mask table
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int p = pbase + 16 * j + l16;
                const int n = n0 + G::p_img(p);
                poff[j] = (unsigned)((((size_t)(n < a.N ? n : 0) * (TH * TW) + G::p_oy(p) * TW + G::p_ox(p)) * a.Cout + chw) >> 3);
                if constexpr (MSKS) mrow[j] = ((a.site.cnt0 + a.t0 + n / a.B) % a.site.num_masks) * a.Cout + chw;     // Masksembles2D: mask (cnt0 + t) mod M
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < TP; ++j) rres[h][j] = *(const half8_e*)(a.res + ((size_t)poff[j] << 3) + 32 * h);
            // the site's Philox calls while the residual is in flight: one call = the 64 channels of (pixel, wave); this lane draws the calls of
            // its own pixel column for the tiles j = kq and kq + 4, the four lanes of a column exchange words (epilogue_lite's distribution)
            philox4 mine[TP / 4];
            if constexpr (MC) {
#pragma unroll
                for (int r = 0; r < TP / 4; ++r) {
                    const int p = pbase + 16 * (kq + 4 * r) + l16;
                    const int n = n0 + G::p_img(p);
                    const int tl = n / a.B;
                    const uint64_t e0 = (uint64_t)((n - tl * a.B) * (TH * TW) + G::p_oy(p) * TW + G::p_ox(p)) * a.Cout + ch0 + 128 * g + wc * 64;
                    mine[r] = philox_site_call(a.site, e0, (uint32_t)(a.t0 + tl));
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4_e sc[2], bi[2];
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    sc[ii] = *(const f32x4_e*)(bn_scale + chw + 32 * h + 4 * ii);
                    bi[ii] = *(const f32x4_e*)(bn_bias + chw + 32 * h + 4 * ii);
                }
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    uint32_t fields = 0u;
                    if constexpr (MC) {
                        const int src = l16 + 16 * (j & 3);
                        const uint32_t wa = (uint32_t)__shfl((int)mine[j >> 2].w[2 * h], src, 64), wb = (uint32_t)__shfl((int)mine[j >> 2].w[2 * h + 1], src, 64);
                        fields = ((kq >> 1) ? wb : wa) >> (16 * (kq & 1));          // channels 32 h + 8 kq ..: word 2 h + (kq >> 1), half kq & 1
                    }
                    half8_e o;
                    f32x4_e pv[2];
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[2 * h + ii][j][e] * sc[ii][e] + bi[ii][e];
                            asm("" : "+v"(v));                                  // (epilogue_lite rounds BN into the accumulator before the residual is added)
                            v += a16_to_f32<BF>(rres[h][j][4 * ii + e]);
                            v = fmaxf(v, 0.f);
                            if constexpr (MC) v = ((fields >> (2 * (4 * ii + e))) & 3u) >= a.site.thresh ? v * a.site.scale : 0.f;
                            if constexpr (MSKS) { const float mk = a.site.masks[mrow[j] + 32 * h + 4 * ii + e]; v = mk == 0.f ? 0.f : v * mk; }
                            asm("" : "+v"(v));                                  // keep the fp32 product (epilogue_lite): rounded once
                            if constexpr (POOLP) {
                                float x = fmaxf(v, 0.f);                       // (the head's ReLU)
                                x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));    // lane ^ 1
                                x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));    // lane ^ 2
                                x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));   // 7 - lane (half row)
                                x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));   // 15 - lane (row)
                                pv[ii][e] = x * (1.f / 16.f);
                            } else {
                                o[4 * ii + e] = a16_from_f32<BF>(v);
                            }
                        }
                    const int p = pbase + 16 * j + l16;
                    const int nimg = n0 + G::p_img(p);
                    if constexpr (POOLP) {
                        if (l16 == 0 && nimg < a.N) {
                            *(f32x4_e*)(a.pool + (size_t)nimg * a.Cout + chw + 32 * h) = pv[0];
                            *(f32x4_e*)(a.pool + (size_t)nimg * a.Cout + chw + 32 * h + 4) = pv[1];
                        }
                    } else if (nimg < a.N) {
                        *(half8_e*)(a.out + ((size_t)poff[j] << 3) + 32 * h) = o;
                    }
                }
            }
        } else if constexpr (PWP_DIRECT) {
            // ---- epilogue straight from the registers: lane (kq, l16) holds, for pixel tile j, channels 8 kq .. + 7 (tiles 0, 1) and
            //      32 + 8 kq .. + 7 (tiles 2, 3) of the wave's 64 channels of pixel pbase + 16 j + l16 ----
            const int chw = ch0 + 128 * g + wc * 64 + 8 * kq;
            f32x4_e sc[4], bi[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c4 = chw + 32 * (i >> 1) + 4 * (i & 1);
                sc[i] = *(const f32x4_e*)(bn_scale + c4);
                bi[i] = *(const f32x4_e*)(bn_bias + c4);
            }
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int p = pbase + 16 * j + l16;
                const int n = n0 + G::p_img(p);
                _Float16* const dst = a.out + ((size_t)n * (TH * TW) + G::p_oy(p) * TW + G::p_ox(p)) * a.Cout + chw;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    half8_e o;
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[2 * h + ii][j][e] * sc[2 * h + ii][e] + bi[2 * h + ii][e];
                            if (a.relu) v = fmaxf(v, 0.f);
                            o[4 * ii + e] = a16_from_f32<BF>(v);
                        }
                    if (n < a.N) *(half8_e*)(dst + 32 * h) = o;
                }
            }
        } else {
            char* const E = smem + L::E_OFF + g * 32768;
            int tl = tid & 255;
            asm volatile("" : "+v"(tl));
            const int chl = ch0 + 128 * g;
            const int k = tl & 15;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                if (rr) lds_barrier();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c4 = chl + wc * 64 + 16 * i + 4 * kq;
                    const f32x4_e sc = *(const f32x4_e*)(bn_scale + c4), bi = *(const f32x4_e*)(bn_bias + c4);
                    const int cq = wc * 8 + 2 * i + (kq >> 1);
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int p = wp * 64 + jj * 16 + l16;
                        half4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[i][4 * rr + jj][e] * sc[e] + bi[e];
                            if (a.relu) v = fmaxf(v, 0.f);
                            o[e] = a16_from_f32<BF>(v);
                        }
                        *(half4*)(E + p * 256 + ((cq ^ l16) << 4) + (((kq ^ jj) & 1) << 3)) = o;
                    }
                }
                lds_barrier();
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    half8_e o8[4];
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int pl = (tl >> 4) + 16 * (4 * hb + it);
                        o8[it] = *(const half8_e*)(E + pl * 256 + ((k ^ (pl & 15)) << 4));
                    }
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int pl = (tl >> 4) + 16 * (4 * hb + it);
                        const int p = (pl >> 6) * 128 + rr * 64 + (pl & 63);
                        const int n = n0 + G::p_img(p);
                        if (n >= a.N) continue;
                        half8_e v = o8[it];
                        if (it & 1) v = __builtin_shufflevector(v, v, 4, 5, 6, 7, 0, 1, 2, 3);
                        *(half8_e*)(a.out + ((size_t)n * (TH * TW) + G::p_oy(p) * TW + G::p_ox(p)) * a.Cout + chl + 8 * k) = v;
                    }
                }
            }
            lds_barrier();      // the staging area (W2 | P1) is free again for the next tile's K-steps
        }
        stores16 = !POOLP && n0 + IMGS <= a.N;          // (a pooled tail: lane 0 of a row stores 32 quads, the others nothing: the next tile drains)
        if (!more) break;
        vb = nvb; ch0 = ch0n; n0 = n0n;
        TILE_RSRC_W(rs_w, ch0);
    }
#undef PWP_STEP
#undef END_OF_STEP_WAIT
#undef WAIT_VM
#undef RAW_BARRIER
#undef ISSUE_P
#undef TILE_RSRC_IN
#undef TILE_RSRC_W
#undef ISSUE_W
#undef A_TILE
#undef BLDS16
}


// Shapes this kernel takes: 3x3 / stride 1 / pad 1 on 8x8 or 4x4 maps with Cout % 256 == 0.
bool conv_takes_pw_kernel(int ksize, int stride, int pad, int cin, int cout, int ho, int wo) {
    return ksize == 3 && stride == 1 && pad == 1 && cin % 64 == 0 && cout % 256 == 0 && ho == wo && (ho == 8 || ho == 4);
}

template <int TW>
static int launch_pw(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    a.xcd_split = xcd_split_for(a.Cout / 256, (size_t)a.Cout * 9 * a.Cin * 2);
    const long tiles = (long)((a.N + PwGeom<TW>::IMGS - 1) / PwGeom<TW>::IMGS) * (a.Cout / 256);
    if (tiles <= 0 || tiles > 0x7fffffffL) return BMI_ERR_INVALID;
    const dim3 grid((unsigned)tiles), block(512);
    const int epi = opt_epilogue_lite() ? conv_epilogue_kind(a, 16) : (conv_epilogue_is_plain(a) ? BMI_EPI_PLAIN : BMI_EPI_GENERAL);
    const int epi_fine = opt_epilogue_lite() && !a.imap && opt_conv_pw() < 3 ? conv_epilogue_kind_launch(a, 16) : epi;   // (the specialised lite forms: same bits)
    if (a.pool) {   // fp32 means over the 4x4 map instead of the map (the conv feeds one exit head only): the lite epilogue on the registers
        if constexpr (TW == 4) {
            if (epi == BMI_EPI_GENERAL) return BMI_ERR_UNSUPPORTED;
            if (PWP_DIRECT && opt_pw_persist() && opt_conv_pw() < 3 && opt_epilogue_lite() == 1 && epi == BMI_EPI_LITE && a.res && a.res_mod >= a.N && a.relu &&
                a.site.kind == BMI_SITE_NONE && !a.imap && !a.in2 && a.Cin % 64 == 0 && a.Cout <= 512 && a.in_mod >= a.N &&
                (size_t)a.H * a.W * a.Cin * PwGeom<4>::IMGS * 2 < 0xfffffff0ull && (size_t)a.N * a.Ho * a.Wo * a.Cout < (8ull << 32)) {
                // the pooled BasicBlock tail on the persistent walk (conv3x3_pwp_kernel<4, .., LITE_RES, POOLP>): the same bits
                static const int n_cu_p = [] {
                    int dev = 0, cu = 0;
                    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cu = 0;
                    return cu > 0 ? cu : 256;
                }();
                const dim3 pgrid((unsigned)(tiles < n_cu_p ? tiles : n_cu_p));
                if (a.bf16) hipLaunchKernelGGL((conv3x3_pwp_kernel<4, true, false, BMI_EPI_LITE_RES, true>), pgrid, block, 0, s, a, (int)tiles);
                else hipLaunchKernelGGL((conv3x3_pwp_kernel<4, false, false, BMI_EPI_LITE_RES, true>), pgrid, block, 0, s, a, (int)tiles);
                BMI_CHECK_LAUNCH();
                return BMI_OK;
            }
            if (a.imap) {
                if (a.bf16) hipLaunchKernelGGL((conv3x3_pw_kernel<4, BMI_EPI_LITE, true, true, true>), grid, block, 0, s, a);
                else hipLaunchKernelGGL((conv3x3_pw_kernel<4, BMI_EPI_LITE, false, true, true>), grid, block, 0, s, a);
            } else if (a.bf16) hipLaunchKernelGGL((conv3x3_pw_kernel<4, BMI_EPI_LITE, true, false, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((conv3x3_pw_kernel<4, BMI_EPI_LITE, false, false, true>), grid, block, 0, s, a);
            BMI_CHECK_LAUNCH();
            return BMI_OK;
        } else {
            return BMI_ERR_UNSUPPORTED;
        }
    }
    if (opt_pw_persist() && opt_conv_pw() < 3 && epi == BMI_EPI_PLAIN && !a.imap && a.Cin % 64 == 0 && a.Cout <= 512 && (!a.in2 || a.Cin2 % 32 == 0) &&
        a.in_mod >= a.N /* a tile's images are consecutive tensor rows */ && (size_t)a.H * a.W * a.Cin * PwGeom<TW>::IMGS * 2 < 0xfffffff0ull) {
        // the persistent form (conv3x3_pwp): one workgroup per CU walks the tiles; the same bits
        static const int n_cu = [] {
            int dev = 0, cu = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cu = 0;
            return cu > 0 ? cu : 256;
        }();
        const dim3 pgrid((unsigned)(tiles < n_cu ? tiles : n_cu));
        if (a.in2) {
            if (a.bf16) hipLaunchKernelGGL((conv3x3_pwp_kernel<TW, true, true>), pgrid, block, 0, s, a, (int)tiles);
            else hipLaunchKernelGGL((conv3x3_pwp_kernel<TW, false, true>), pgrid, block, 0, s, a, (int)tiles);
        } else {
            if (a.bf16) hipLaunchKernelGGL((conv3x3_pwp_kernel<TW, true, false>), pgrid, block, 0, s, a, (int)tiles);
            else hipLaunchKernelGGL((conv3x3_pwp_kernel<TW, false, false>), pgrid, block, 0, s, a, (int)tiles);
        }
        BMI_CHECK_LAUNCH();
        return BMI_OK;
    }
    if (opt_conv_pw() >= 3 && !a.in2 && !a.imap && !a.bf16 && epi != BMI_EPI_GENERAL && a.Cin % 64 == 0) {   // the four-wave form ("conv_pw" = 3 | 4)
        const dim3 block4(256);
        if (epi == BMI_EPI_PLAIN) hipLaunchKernelGGL((conv3x3_pw4_kernel<TW, BMI_EPI_PLAIN, false, false>), grid, block4, 0, s, a);
        else hipLaunchKernelGGL((conv3x3_pw4_kernel<TW, BMI_EPI_LITE, false, false>), grid, block4, 0, s, a);
        BMI_CHECK_LAUNCH();
        return BMI_OK;
    }
    if (PWP_DIRECT && opt_pw_persist() && opt_conv_pw() < 3 && (epi_fine == BMI_EPI_LITE_RES || epi_fine == BMI_EPI_LITE_RES_MC || epi_fine == BMI_EPI_LITE_RES_MSK) && !a.in2 && a.Cin % 64 == 0 &&
        a.Cout <= 512 && a.in_mod >= a.N && (size_t)a.H * a.W * a.Cin * PwGeom<TW>::IMGS * 2 < 0xfffffff0ull && (size_t)a.N * a.Ho * a.Wo * a.Cout < (8ull << 32)) {
        // the BasicBlock tails on the persistent walk, finished straight from the registers (conv3x3_pwp_kernel<.., EPIK>): the same bits
        static const int n_cu = [] {
            int dev = 0, cu = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cu = 0;
            return cu > 0 ? cu : 256;
        }();
        const dim3 pgrid((unsigned)(tiles < n_cu ? tiles : n_cu));
        if (epi_fine == BMI_EPI_LITE_RES) {
            if (a.bf16) hipLaunchKernelGGL((conv3x3_pwp_kernel<TW, true, false, BMI_EPI_LITE_RES>), pgrid, block, 0, s, a, (int)tiles);
            else hipLaunchKernelGGL((conv3x3_pwp_kernel<TW, false, false, BMI_EPI_LITE_RES>), pgrid, block, 0, s, a, (int)tiles);
        } else if (epi_fine == BMI_EPI_LITE_RES_MSK) {
            if (a.bf16) hipLaunchKernelGGL((conv3x3_pwp_kernel<TW, true, false, BMI_EPI_LITE_RES_MSK>), pgrid, block, 0, s, a, (int)tiles);
            else hipLaunchKernelGGL((conv3x3_pwp_kernel<TW, false, false, BMI_EPI_LITE_RES_MSK>), pgrid, block, 0, s, a, (int)tiles);
        } else {
            if (a.bf16) hipLaunchKernelGGL((conv3x3_pwp_kernel<TW, true, false, BMI_EPI_LITE_RES_MC>), pgrid, block, 0, s, a, (int)tiles);
            else hipLaunchKernelGGL((conv3x3_pwp_kernel<TW, false, false, BMI_EPI_LITE_RES_MC>), pgrid, block, 0, s, a, (int)tiles);
        }
        BMI_CHECK_LAUNCH();
        return BMI_OK;
    }
    if (epi_fine == BMI_EPI_LITE_RES || epi_fine == BMI_EPI_LITE_RES_MC || epi_fine == BMI_EPI_LITE_RES_MSK) {
        if (epi_fine == BMI_EPI_LITE_RES) {
            if (a.bf16) hipLaunchKernelGGL((conv3x3_pw_kernel<TW, BMI_EPI_LITE_RES, true, false>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((conv3x3_pw_kernel<TW, BMI_EPI_LITE_RES, false, false>), grid, block, 0, s, a);
        } else if (epi_fine == BMI_EPI_LITE_RES_MSK) {
            if (a.bf16) hipLaunchKernelGGL((conv3x3_pw_kernel<TW, BMI_EPI_LITE_RES_MSK, true, false>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((conv3x3_pw_kernel<TW, BMI_EPI_LITE_RES_MSK, false, false>), grid, block, 0, s, a);
        } else {
            if (a.bf16) hipLaunchKernelGGL((conv3x3_pw_kernel<TW, BMI_EPI_LITE_RES_MC, true, false>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((conv3x3_pw_kernel<TW, BMI_EPI_LITE_RES_MC, false, false>), grid, block, 0, s, a);
        }
        BMI_CHECK_LAUNCH();
        return BMI_OK;
    }
#define PW_LAUNCH_BF(BF_, IMAP_)                                                                                                    \
    {                                                                                                                               \
        if (epi == BMI_EPI_PLAIN) hipLaunchKernelGGL((conv3x3_pw_kernel<TW, BMI_EPI_PLAIN, BF_, IMAP_>), grid, block, 0, s, a);     \
        else if (epi == BMI_EPI_LITE) hipLaunchKernelGGL((conv3x3_pw_kernel<TW, BMI_EPI_LITE, BF_, IMAP_>), grid, block, 0, s, a);  \
        else hipLaunchKernelGGL((conv3x3_pw_kernel<TW, BMI_EPI_GENERAL, BF_, IMAP_>), grid, block, 0, s, a);                        \
    }
#define PW_LAUNCH(IMAP_) \
    if (a.bf16) PW_LAUNCH_BF(true, IMAP_) else PW_LAUNCH_BF(false, IMAP_)
    if (a.imap) { PW_LAUNCH(true) } else { PW_LAUNCH(false) }
#undef PW_LAUNCH
#undef PW_LAUNCH_BF
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}

// BMI_ERR_UNSUPPORTED -> the caller falls back to conv3x3_patch.
// Minimum-grid rule: a grid that cannot fill 3/4 of the CUs is better served by conv3x3_patch's 128 x 128 tiles (VGG-11's
// deterministic 512-channel convs on B = 250 images: 16-63 tiles; 0.061 vs 0.029 ms per launch).  The rule looks at the
// engine's FULL-CHUNK image count (ConvArgs::n_ref: B x planned chunk for the sample-folded suffix, B for the prefix), never
// at how many samples this launch carries: a t-shard (one rank of eight, or the last partial chunk) runs the same kernel as
// the single-rank run and gets the same bits (conv3x3_patch sums the channels in 64-wide chunks, this kernel in 32-wide
// ones: equal to fp32 rounding, not bit for bit).  "conv_pw" = 2 drops the rule (tests).
int launch_conv3x3_pw(const ConvArgs& a, hipStream_t s) {
    if (!opt_conv_pw() || a.in_bits || a.wgt_b) return BMI_ERR_UNSUPPORTED;
    if (opt_conv_pw() != 2 && opt_conv_pw() != 4) {
        static const int n_cu = [] {
            int dev = 0, cu = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cu = 0;
            return cu;
        }();
        const int imgs = a.Ho == 8 ? 4 : 16, n_sel = a.n_ref > 0 ? a.n_ref : a.N;
        if ((long)((n_sel + imgs - 1) / imgs) * (a.Cout / 256) < (n_cu > 0 ? 3 * n_cu / 4 : 192)) return BMI_ERR_UNSUPPORTED;
    }
    if (a.in2 && (!a.wgt2 || a.Cin2 % 64 != 0 || a.in2_mod <= 0 || a.stride2 < 1)) return BMI_ERR_INVALID;
    if (!conv_takes_pw_kernel(a.ksize, a.stride, a.pad, a.Cin, a.Cout, a.Ho, a.Wo)) return BMI_ERR_UNSUPPORTED;
    if (a.N <= 0 || a.in_mod <= 0 || a.B <= 0 || (a.res && a.res_mod <= 0)) return BMI_ERR_INVALID;
    if ((size_t)a.in_mod * a.H * a.W * a.Cin >= 0x7fffffffull) return BMI_ERR_UNSUPPORTED;   // 31-bit DMA source offsets
    if (a.in2 && (size_t)a.in2_mod * a.H2 * a.W2 * a.Cin2 >= 0x7fffffffull) return BMI_ERR_UNSUPPORTED;
    return a.Ho == 8 ? launch_pw<8>(a, s) : launch_pw<4>(a, s);
}
