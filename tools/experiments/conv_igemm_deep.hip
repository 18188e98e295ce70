// Per-tap implicit-GEMM convolution with a DEEP LDS-DMA pipeline: conv_igemm_wide's 256 x 256 tile, 8 waves and ping-pong
// main loop, with the K-step cut to 32 channels and FOUR [weights | pixels] stages of 32 KB instead of two of 64 KB.
//
// Why.  conv_igemm_wide waits vmcnt(0) once per 64-deep K-step for tiles it requested one K-step (~2000 cycles) earlier; its
// activation tile comes from HBM / beyond L2 (first touch), and the PMC wave-cycle split shows 43 % of the wave cycles
// waiting.  conv3x3_pw showed what the distance is worth: the same loop went from 845-1030 to 1245-1400 TFLOP/s when its
// weight tile was requested two K-steps ahead instead of one.  Here the tiles of K-step T+3 are requested during K-step T
// (~3000 cycles ahead) at the same 128 KB of LDS, and the per-step wait is a COUNTED vmcnt that leaves the two younger
// stages in flight.
//
//   stage      = [256 channel rows | 256 pixel rows] x 32 k (64-byte rows); 16-byte chunk c of row r is stored at position
//                (c + (r >> 2)) & 3 (four rows share a 256-byte bank window: conflict-free ds_read_b128, see conv3x3_pw.hip);
//                the DMA writes lane-linearly, so the rotation is applied to the per-lane SOURCE address
//   K-step     = one tap x 32 channels = one v_mfma_f32_16x16x32 per tile; per thread 2 + 2 DMA instructions
//   main loop  = two phases per K-step (LOAD part, barrier, MFMA part, barrier), the two wave groups one barrier apart;
//                stage of step T = T & 3.  WAR: stage (T+3)&3 was last read in step T-1 (group 1's phase-1 LOAD, interval
//                4T-1, retired by its lgkmcnt(0) in 4T) and is refilled from interval 4T+1 (group 1) / 4T+2 (group 0).  RAW: in interval 4T+3 every
//                wave waits until only the DMA of steps T+2 and T+3 (8 instructions) is in flight, i.e. until its pieces of
//                step T+1 have landed, before the barrier that ends 4T+3; first reads of step T+1 in 4T+4.
//   everything else (tile order, pair mode, epilogue per channel half, accumulator layout) is conv_igemm_wide's.
// Reference semantics: the stride-2 3x3 convs of BasicBlock / the exit heads (SA/models/resnet18/resnet18.py:280-299,
// :306-329) and the 1x1 convs of Bottleneck nets (:51-85).
#include <cstdlib>

#include "conv_epilogue.h"
#include "kernels.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

static __device__ unsigned int g_zero_page_d[64];

#define GLDS16(SRC, LDSPTR)                                                                     \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC),       \
                                     (__attribute__((address_space(3))) void*)(LDSPTR), 16, 0, 0)

#define DST 32768   // bytes of one stage
#define DNST 4

template <bool PLAIN, bool BF, bool IMAP>
__global__ __launch_bounds__(512, 1) void conv_igemm_deep_kernel(ConvArgs a) {
    constexpr int TJ = 4, TI = 4, TP = 8;
    typedef float accv __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) char smem[DNST * DST];
    static_assert(DNST * DST == 2 * BMI_EPILOGUE_LDS_BYTES, "one epilogue staging area per channel half");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, kq = lane >> 4;
    const int g = wave >> 2, wc = (wave >> 1) & 1, wp = wave & 1;

    const int n_ctiles = a.Cout / 256;
    int ptile, ctile;
    xcd_tile_map(blockIdx.x, (a.M + 255) / 256, n_ctiles, ptile, ctile, a.xcd_split);
    const int ch0 = ctile * 256;
    const int pix0 = ptile * 256;
    const int HoWo = a.Ho * a.Wo;
    const int Ktot = a.ksize * a.ksize * a.Cin;
    const int split = a.wgt_b ? a.split : a.Cout;

    // ---- per-thread DMA sources: piece q = tid + 512 i -> tile row (tid >> 2) + 128 i, position tid & 3 ----
    const int lg = (((tid & 3) - (tid >> 4)) & 3) * 8;            // logical k-chunk held at this position ((row >> 2) & 3 == (tid >> 4) & 3)
    const _Float16* wsrc[2];
    const _Float16* xsrc[2];
    int iy0[2], ix0[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ch = ch0 + (tid >> 2) + 128 * i;
        wsrc[i] = (ch < split ? a.wgt + (size_t)ch * Ktot : a.wgt_b + (size_t)(ch - split) * Ktot) + lg;
        const int m = pix0 + (tid >> 2) + 128 * i;
        const bool vm = m < a.M;
        const int mm = vm ? m : 0;
        const int n = mm / HoWo;
        const int rem = mm - n * HoWo;
        const int oy = rem / a.Wo;
        const int ox = rem - oy * a.Wo;
        iy0[i] = vm ? oy * a.stride - a.pad : -0x10000;          // a row beyond M never passes the bounds test
        ix0[i] = ox * a.stride - a.pad;
        xsrc[i] = a.in + (size_t)(map_image<IMAP>(a, n) % a.in_mod) * a.H * a.W * a.Cin + lg;
    }
    // K-step T = (tap, 32-channel chunk): DMA of its two tiles into stage T & 3
#define ISSUE_STEP(KY, KX, C0, ST)                                                                     \
    {                                                                                                  \
        const int koff_ = ((KY) * a.ksize + (KX)) * a.Cin + (C0);                                      \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                  \
            GLDS16(wsrc[i] + koff_, smem + (ST) * DST + (i * 512 + wave * 64) * 16);                   \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                \
            const int iy_ = iy0[i] + (KY), ix_ = ix0[i] + (KX);                                        \
            const bool ok_ = (unsigned)iy_ < (unsigned)a.H && (unsigned)ix_ < (unsigned)a.W;           \
            const _Float16* s_ = ok_ ? xsrc[i] + (size_t)(iy_ * a.W + ix_) * a.Cin + (C0)              \
                                     : (const _Float16*)g_zero_page_d;                                 \
            GLDS16(s_, smem + (ST) * DST + 16384 + (i * 512 + wave * 64) * 16);                        \
        }                                                                                              \
    }
#define NEXT_K(KY, KX, C0)                                                                             \
    {                                                                                                  \
        (C0) += 32;                                                                                    \
        if ((C0) == a.Cin) {                                                                           \
            (C0) = 0;                                                                                  \
            if (++(KX) == a.ksize) { (KX) = 0; ++(KY); }                                               \
        }                                                                                              \
    }
    const int nK = a.ksize * a.ksize * (a.Cin / 32);
    int ky = 0, kx = 0, c0 = 0;             // the next K-step to be requested
    ISSUE_STEP(ky, kx, c0, 0);
    NEXT_K(ky, kx, c0);
    if (nK > 1) { ISSUE_STEP(ky, kx, c0, 1); NEXT_K(ky, kx, c0); }
    if (nK > 2) { ISSUE_STEP(ky, kx, c0, 2); NEXT_K(ky, kx, c0); }
    __builtin_amdgcn_sched_barrier(0);

    accv acc[TI][TP];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

    const int fbyte = ((kq + (l16 >> 2)) & 3) << 4;                // position of chunk kq in this lane's rows
    const int a_off = (g * 128 + wc * 64 + l16) * 64 + fbyte;
    const int b_off = 16384 + (wp * 128 + l16) * 64 + fbyte;

#define RAW_BARRIER()                                  \
    {                                                  \
        __builtin_amdgcn_sched_barrier(0);             \
        asm volatile("" ::: "memory");                 \
        __builtin_amdgcn_s_barrier();                  \
        asm volatile("" ::: "memory");                 \
        __builtin_amdgcn_sched_barrier(0);             \
    }
    // stage 0 must have landed; the 8 instructions of stages 1 and 2 may fly on
    if (nK > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nK > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RAW_BARRIER();
    if (g == 1) RAW_BARRIER();          // stagger
    half8 af[TI], bf[4];
    const int dma_phase = g == 0 ? 1 : 0;   // group 1 reads the stage being refilled last (its phase-1 LOAD of step T-1, interval 4T-1,
                                            // retired in 4T): group 1 requests from 4T+1 (its phase 0), group 0 from 4T+2 (its phase 1)
    for (int ks = 0; ks < nK; ++ks) {
        const char* st = smem + (ks & 3) * DST;
        const bool req = ks + 3 < nK;     // request K-step ks + 3 into stage (ks + 3) & 3
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            // LOAD part
            if (kk == 0) {
#pragma unroll
                for (int i = 0; i < TI; ++i) af[i] = *(const half8*)(st + a_off + i * 16 * 64);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = *(const half8*)(st + b_off + (4 * kk + j) * 16 * 64);
            if (kk == dma_phase && req) {
                ISSUE_STEP(ky, kx, c0, (ks + 3) & 3);
                NEXT_K(ky, kx, c0);
            }
            if (kk == 1 && g == 1) {      // interval 4T+3, group 1 (LOAD part): its pieces of step T+1 have landed
                const int left = nK - 1 - ks;
                if (left >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (left == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            RAW_BARRIER();
            // MFMA part
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][4 * kk + j] = mfma_16x16x32<BF>(af[i], bf[j], acc[i][4 * kk + j]);
            __builtin_amdgcn_s_setprio(0);
            if (kk == 1 && g == 0) {      // interval 4T+3, group 0 (MFMA part)
                const int left = nK - 1 - ks;
                if (left >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (left == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            RAW_BARRIER();
        }
    }
    if (g == 0) RAW_BARRIER();          // re-align the two groups before the epilogue reuses the LDS
#undef RAW_BARRIER
#undef ISSUE_STEP
#undef NEXT_K
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: each channel half (4 waves) in its own 64 KB ----
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
    epilogue_coalesced<TJ, PLAIN, 16, BF>(b, acc, smem + g * BMI_EPILOGUE_LDS_BYTES, tid & 255, chg, pixmap, offmap);
}

// Same shapes and call contract as launch_conv_igemm_wide (which calls this when "wide_deep" is on).
int launch_conv_igemm_deep(const ConvArgs& a, long blocks, hipStream_t s) {
    const dim3 grid((unsigned)blocks), block(512);
    const bool plain = conv_epilogue_is_plain(a);
#define DEEP_LAUNCH(IMAP_)                                                                                            \
    if (a.bf16) {                                                                                                     \
        if (plain) hipLaunchKernelGGL((conv_igemm_deep_kernel<true, true, IMAP_>), grid, block, 0, s, a);             \
        else hipLaunchKernelGGL((conv_igemm_deep_kernel<false, true, IMAP_>), grid, block, 0, s, a);                  \
    } else {                                                                                                          \
        if (plain) hipLaunchKernelGGL((conv_igemm_deep_kernel<true, false, IMAP_>), grid, block, 0, s, a);            \
        else hipLaunchKernelGGL((conv_igemm_deep_kernel<false, false, IMAP_>), grid, block, 0, s, a);                 \
    }
    if (a.imap) { DEEP_LAUNCH(true) } else { DEEP_LAUNCH(false) }
#undef DEEP_LAUNCH
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}
